# usage (GPU box): bash tools/envab2.sh -- the headline step under library debug switches (MESHVAE_DEBUG), 200-step runs, one box
run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us')"; }
run A=0
run MESHVAE_DEBUG=fork_batch=2
run MESHVAE_DEBUG=fork_batch=4
run A=0
run MESHVAE_DEBUG=dw_lane2=1
run MESHVAE_DEBUG=no_dx_first=1
run MESHVAE_DEBUG=fork_batch=3
run A=0
