#!/bin/bash
# usage (on the GPU box): tools/timeline.sh <tag> [bench args] -- kernel timeline of one step to gpurun_out/<tag>_timeline.txt
TAG="$1"; shift
export TMPDIR=/tmp
OUT="gpurun_out/tl_$TAG"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 bench.py --steps 40 --warmup 10 --prewarm-steps 0 --no-cpu-baseline --no-kernel-roofline --no-variants "$@" > "gpurun_out/${TAG}_timeline.log" 2>&1
python tools/timeline_csv.py "$OUT" > "gpurun_out/${TAG}_timeline.txt" 2>&1
rm -rf "$OUT"
