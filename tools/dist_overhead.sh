#!/bin/bash
# Where does the multi-rank launch path lose time on ONE GPU?  A: plain, B: 1-rank RCCL group in-process,
# C: under torch.distributed.run without a group, D: torch.distributed.run + 1-rank group.
ARGS="--steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --prewarm-steps 300"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['ms_per_step'],4), 'ms', d['final_loss'])" "$1" "$2"; }
python bench.py $ARGS > gpurun_out/do_a.json 2>/dev/null && pick gpurun_out/do_a.json A_plain &&
python bench.py --rehearse-allreduce $ARGS > gpurun_out/do_b.json 2>/dev/null && pick gpurun_out/do_b.json B_group_inproc &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 $ARGS > gpurun_out/do_c.json 2>/dev/null && pick gpurun_out/do_c.json C_torchrun_nogroup &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --rehearse-allreduce $ARGS > gpurun_out/do_d.json 2>/dev/null && pick gpurun_out/do_d.json D_torchrun_group &&
OMP_NUM_THREADS=1 python bench.py $ARGS > gpurun_out/do_e.json 2>/dev/null && pick gpurun_out/do_e.json E_plain_omp1
