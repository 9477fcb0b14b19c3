# usage (GPU box): bash tools/envabn.sh "sw=a" "sw=b,sw2=c" ... -- rounds of the headline step, one run per debug-switch setting ("-" = none)
run() { env MESHVAE_DEBUG="$1" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants ${BENCH_EXTRA} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step']*1e3,1), 'us', d['final_loss'])"; }
for i in 1 2 3; do for s in "$@"; do [ "$s" = "-" ] && s=""; run "$s"; done; done
