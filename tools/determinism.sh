#!/bin/bash
# run-to-run determinism of the three training configurations (1510 timed steps each, twice): ms_per_step and the final loss
# must repeat bit for bit (fixed-order reductions everywhere).  -> profiles/<round>_determinism_1510.txt
for cfg in "" "--dtype bf16" "--config hires20k"; do for i in 1 2; do
  python bench.py --steps 1500 --warmup 10 --prewarm-steps 0 --no-cpu-baseline --no-kernel-roofline --no-variants $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], repr(d['final_loss']))"
done; done
echo "# the step over the batch size (fp32, 300 steps each)"
for b in 1 8 16 32 48 56 64 65 128 256; do python bench.py --batch $b --steps 300 --warmup 20 --prewarm-steps 100 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B = $b', round(d['ms_per_step']*1e3,1), 'us', round(d['value']), 'meshes/s')"; done
