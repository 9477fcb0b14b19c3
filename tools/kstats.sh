#!/bin/bash
# usage (on the GPU box): tools/kstats.sh <tag> [bench args] -- rocprofv3 kernel stats of a short bench run, top kernels
# to gpurun_out/<tag>_kstats.txt (raw trace deleted)
TAG="$1"; shift
export TMPDIR=/tmp
OUT="gpurun_out/ks_$TAG"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 bench.py --steps 50 --warmup 10 --prewarm-steps 0 --no-cpu-baseline --no-kernel-roofline --no-variants "$@" > "gpurun_out/${TAG}_kstats.log" 2>&1
f=$(find "$OUT" -name "s_kernel_stats.csv" | head -1)
python - "$f" > "gpurun_out/${TAG}_kstats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print(f"{float(r['AverageNs'])/1e3:9.2f} us x{r['Calls']:>6} {float(r['Percentage']):6.2f}%  {r['Name'][:110]}")
PY
rm -rf "$OUT"
