#!/bin/bash
# usage (on the GPU box): tools/roctx_profile.sh <tag> [bench args] -- kernel + marker trace of a few steps with the per-layer
# roctx ranges of the step engine on (debug switch roctx): gpurun_out/<tag>_roctx_{kernel,marker}.csv
TAG="$1"; shift
export TMPDIR=/tmp
OUT="gpurun_out/roctx_$TAG"
MESHVAE_DEBUG=roctx=1 rocprofv3 --kernel-trace --marker-trace --output-format csv -d "$OUT" -o t -- python3 bench.py --steps 10 --warmup 5 --prewarm-steps 0 --no-cpu-baseline --no-kernel-roofline --no-variants "$@" > "gpurun_out/${TAG}_roctx.log" 2>&1
find "$OUT" -name "*marker*csv" -exec cp {} "gpurun_out/${TAG}_roctx_marker.csv" \;
find "$OUT" -name "*kernel_trace*csv" -exec cp {} "gpurun_out/${TAG}_roctx_kernel.csv" \;
rm -rf "$OUT"
