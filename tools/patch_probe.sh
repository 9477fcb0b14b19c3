#!/bin/bash
# Patch kernels (csrc/cheb_patch.hip) against the slab kernels they replace, one 16 -> 16 layer of the 5k level at B = 64:
# isolated launches through the C ABI (tools/microbench_conv.py).  Output: gpurun_out/r05_patch_probe.txt
set -o pipefail
out=gpurun_out/r05_patch_probe.txt
: > $out
run() { echo "## $*" >> $out; env "$@" >> $out 2>&1; }
for mode in "--iters 50" "--iters 50 --bwd" "--iters 50 --dwonly"; do
  run MESHVAE_DEBUG=no_patch=1 timeout -k 10 120 python tools/microbench_conv.py $mode || exit 1
  run MESHVAE_DEBUG=no_patch=0 timeout -k 10 120 python tools/microbench_conv.py $mode || exit 1
done
cat $out
