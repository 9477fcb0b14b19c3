#!/bin/bash
# one bench run per runtime knob (300 steps), plain runs first / middle / last for the box's drift
run() { env "$@" python bench.py --steps 300 --warmup 30 --prewarm-steps 300 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us')" || echo "$* FAILED"; }
run A=0
for e in AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 ROC_SKIP_KERNEL_ARG_COPY=1 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1; do run $e; done
run A=0
for e in DEBUG_HIP_DYNAMIC_QUEUES=1 DEBUG_HIP_DYNAMIC_QUEUES=0 GPU_FORCE_QUEUE_PROFILING=1 ROC_AQL_QUEUE_SIZE=1024 ROC_AQL_QUEUE_SIZE=65536 DEBUG_CLR_MAX_BATCH_SIZE=100 DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1000 GPU_NUM_COMPUTE_RINGS=4 ROC_SYSTEM_SCOPE_SIGNAL=0 ROC_CPU_WAIT_FOR_SIGNAL=0 ROC_ACTIVE_WAIT_TIMEOUT=0 GPU_FLUSH_ON_EXECUTION=1 AMD_SERIALIZE_KERNEL=0; do run $e; done
run A=0
