# usage: tools/scratch/ab20k.sh rounds "<debug A>" "<debug B>" ... -- alternating hires20k bench runs
R="$1"; shift
run() { MESHVAE_DEBUG="$1" timeout -k 10 300 python bench.py --config hires20k --steps 60 --warmup 10 --prewarm-steps 20 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1]', round(d['ms_per_step']*1e3,1), 'us')"; }
for i in $(seq $R); do for cfg in "$@"; do run "$cfg" || exit 1; done; done
