#!/bin/bash
# The 1-rank RCCL rehearsal of bench.py under different hardware-queue limits (bench.py itself defaults the
# variable to 8), then the launcher path the driver uses.
ARGS="--steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --prewarm-steps 300"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['ms_per_step'],4), 'ms', d['final_loss'])" "$1" "$2"; }
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --rehearse-allreduce $ARGS > gpurun_out/dq_$q.json 2>/dev/null && pick gpurun_out/dq_$q.json group_hwq$q || exit 1
done
python bench.py $ARGS > gpurun_out/dq_plain.json 2>/dev/null && pick gpurun_out/dq_plain.json plain_default || exit 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --rehearse-allreduce $ARGS > gpurun_out/dq_tr.json 2>/dev/null && pick gpurun_out/dq_tr.json torchrun_group_default
