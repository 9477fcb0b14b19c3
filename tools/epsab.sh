run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us', d['final_loss'])"; }
for i in 1 2 3; do run MESHVAE_EPS_AHEAD=0; run MESHVAE_EPS_AHEAD=1; done
