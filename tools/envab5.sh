#!/bin/bash
# alternating confirmation of the knobs tools/envab4.sh flagged
run() { env "$@" python bench.py --steps 300 --warmup 30 --prewarm-steps 300 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us')" || echo "$* FAILED"; }
for i in 1 2 3 4; do run A=0; run DEBUG_HIP_KERNARG_COPY_OPT=0; run ROC_USE_FGS_KERNARG=1; run DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_USE_FGS_KERNARG=1; run ROC_CPU_WAIT_FOR_SIGNAL=0; done
