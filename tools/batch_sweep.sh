#!/bin/bash
# the fp32 training step over the batch size (300 timed steps each) -> profiles/<round>_batch_sweep.txt
echo "# python bench.py --batch B --steps 300 --warmup 20 --prewarm-steps 100 (fp32, configs[1] otherwise)"
for b in ${SWEEP_B:-1 8 16 32 48 56 64 65 72 80 96 112 128 192 256}; do python bench.py --batch $b --steps 300 --warmup 20 --prewarm-steps 100 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B = $b', round(d['ms_per_step']*1e3,1), 'us', round(d['value']), 'meshes/s')"; done
