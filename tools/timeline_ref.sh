#!/bin/bash
# usage (on the GPU box): tools/timeline_ref.sh <tag> -- kernel timeline of one step of the reference-API loop
# (bench.py --only-reference-loop) to gpurun_out/<tag>_timeline.txt; MESHVAE_ASYNC from the environment
TAG="$1"; shift
export TMPDIR=/tmp
OUT="gpurun_out/tl_$TAG"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 bench.py --only-reference-loop --steps 40 --warmup 10 --prewarm-steps 0 "$@" > "gpurun_out/${TAG}_timeline.log" 2>&1
python tools/timeline_csv.py "$OUT" > "gpurun_out/${TAG}_timeline.txt" 2>&1
rm -rf "$OUT"
