# usage (GPU box): bash tools/envab.sh -- the headline step under runtime environment knobs (200-step runs, one box)
run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us')"; }
run A=0
run HSA_NO_SCRATCH_RECLAIM=1
run HSA_ENABLE_INTERRUPT=0
run A=0
run GPU_MAX_HW_QUEUES=16
run HSA_NO_SCRATCH_RECLAIM=1 HSA_ENABLE_INTERRUPT=0
run ROC_SIGNAL_POOL_SIZE=4096
run A=0
