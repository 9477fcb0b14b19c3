#!/bin/bash
# 1-rank RCCL rehearsal: plain vs single-collective vs overlapped two-bucket all-reduce (bench.py, one GPU).
ARGS="--steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --prewarm-steps 300"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['ms_per_step'],4), 'ms', d['final_loss'])" "$1" "$2"; }
python bench.py $ARGS > gpurun_out/d3_plain.json 2>/dev/null && pick gpurun_out/d3_plain.json plain || exit 1
python bench.py --rehearse-allreduce $ARGS > gpurun_out/d3_one.json 2>/dev/null && pick gpurun_out/d3_one.json group_one_collective || exit 1
python bench.py --rehearse-allreduce --ar-overlap $ARGS > gpurun_out/d3_two.json 2>gpurun_out/d3_two.err && pick gpurun_out/d3_two.json group_two_buckets_overlapped || { tail -5 gpurun_out/d3_two.err; exit 1; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --rehearse-allreduce --ar-overlap $ARGS > gpurun_out/d3_tr.json 2>/dev/null && pick gpurun_out/d3_tr.json torchrun_two_buckets
