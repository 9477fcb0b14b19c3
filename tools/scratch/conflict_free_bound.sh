for i in 1 2; do
for m in "" "own_rows"; do
  f=$(MESHVAE_PLAN_TIMING_ONLY=$m timeout -k 10 120 python tools/microbench_conv.py --iters 50 2>&1 | tail -1)
  b=$(MESHVAE_PLAN_TIMING_ONLY=$m timeout -k 10 120 python tools/microbench_conv.py --iters 50 --bwd 2>&1 | tail -1)
  echo "[plan lists: ${m:-real}] $f | $b"
done; done
