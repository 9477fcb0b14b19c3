#!/bin/bash
# Runs each tools/graph_diag.py scenario ONCE, in its own process; stops at the first one that hangs (timeout).
OUT=gpurun_out/graph_diag; mkdir -p $OUT
for sc in plain memset_fork nested_fork nested_fork_memset step_nested; do
  timeout -k 10 180 python -X faulthandler tools/graph_diag.py $sc $OUT > $OUT/$sc.log 2>&1
  rc=$?
  echo "scenario $sc rc=$rc: $(grep -E 'scenario|FAILED|Fatal|Segmentation' $OUT/$sc.log | tail -2 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "hung: stopping"; exit 1; fi
done
exit 0
