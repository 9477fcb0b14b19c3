#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench args, e.g. --config hires20k / --dtype bf16 ...]
# 1. plain bench line                    -> gpurun_out/<tag>/bench_n1.json
# 2. rocprofv3 --kernel-trace --stats    -> gpurun_out/<tag>/stats/   (program directly after `--`)
# 3. PMC passes, each in its own run with --kernel-trace only (no --stats-free trace domains):
#    FETCH_SIZE | WRITE_SIZE | SQ (MFMA busy, LDS conflicts, ...) + GRBM_GUI_ACTIVE
# tools/pmc_fold.py turns 2+3 into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json.
set -e -o pipefail
TAG="$1"; shift
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
CONFIG=train5k; DTYPE=f32; prev=""
for a in "$@"; do
  [ "$prev" = "--config" ] && CONFIG="$a"
  [ "$prev" = "--dtype" ] && DTYPE="$a"
  prev="$a"
done
python -c "import bench; print(bench.source_sha())" > "$OUT/src_sha.txt"
export TMPDIR=/tmp
BENCH_ARGS="$@"
python bench.py --steps 100 --warmup 20 $BENCH_ARGS > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench done: $(cut -c1-200 $OUT/bench_n1.json)"
SHORT="--steps 20 --warmup 3 --prewarm-steps 0 --no-cpu-baseline --no-kernel-roofline --no-variants $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --steps 100 --warmup 20 --prewarm-steps 100 --no-cpu-baseline --no-kernel-roofline --no-variants $BENCH_ARGS > "$OUT/stats.log" 2>&1
echo "stats done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o p -- python3 bench.py $SHORT > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $name FAILED (see log)"
  echo "pmc $name done"
done
python tools/pmc_fold.py "$OUT" "$TAG" "$OUT" "$CONFIG" "$DTYPE"
# gpurun copies at most 64 MiB back: keep the folded tables (copy them into profiles/), drop the raw rocprofv3 output
[ -n "$KEEP_RAW" ] || rm -rf "$OUT"/stats "$OUT"/pmc_*/
