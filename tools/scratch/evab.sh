run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us')"; }
run MVH_EXTRA_EVENTS=0; run MVH_EXTRA_EVENTS=20; run MVH_EXTRA_EVENTS=0; run MVH_EXTRA_EVENTS=40; run MVH_EXTRA_EVENTS=0
