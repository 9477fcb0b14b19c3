# usage (GPU box): bash tools/envab3.sh [switch=value] -- alternating runs of the headline step with / without one library debug switch
SW="${1:-no_stop_events=1}"
run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us', d['final_loss'])"; }
for i in 1 2 3 4; do run MESHVAE_DEBUG=$SW; run A=default; done
