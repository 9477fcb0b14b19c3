#!/bin/bash
# tools/fourth_queue.sh -> gpurun_out/r05_fourth_queue.txt : the three modes of tools/fourth_queue_probe.py under the HIP
# default number of hardware queues, 4 (explicit) and 8, each in a fresh process
out=gpurun_out/r05_fourth_queue.txt
: > $out
for q in "" 4 8; do
  for mode in plain surrogate rccl; do
    if [ -z "$q" ]; then env -u GPU_MAX_HW_QUEUES timeout -k 10 200 python tools/fourth_queue_probe.py --mode $mode >> $out 2>&1 || echo "FAILED default $mode" >> $out
    else GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python tools/fourth_queue_probe.py --mode $mode >> $out 2>&1 || echo "FAILED $q $mode" >> $out; fi
  done
done
grep -v amdgpu.ids $out
