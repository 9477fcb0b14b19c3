#!/usr/bin/env python3
"""Does the B = 64 train step survive a FOURTH busy hardware queue -- what a data-parallel rank has beside the step's
three (caller's stream + two weight-gradient lanes): the collective library's own stream?  (VERDICT r4 #2.)

    python tools/fourth_queue_probe.py --mode plain|surrogate|rccl [--steps 1000]

  plain      engine.TrainStep as bench.py runs it (three busy queues)
  surrogate  + one ~30 us kernel per step on an extra stream, ordered by events exactly where engine.py puts the gradient
             all-reduce (after the backward's join, before Adam): record on the step's stream -> wait on the extra stream
             -> kernel -> record -> wait on the step's stream
  rccl       + the real thing with one rank: a 1-rank RCCL group, torch.distributed.all_reduce of the flat gradient
             buffer every step (TrainStep(rehearse_allreduce=True))

Prints ms/step, the per-step host enqueue time distribution and how many steps' enqueue took >= 1 ms (a stalled
hipLaunchKernel shows there: the host is otherwise ~60 us ahead of the GPU).  GPU_MAX_HW_QUEUES is read from the
environment by the HIP runtime: run once per setting."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd")):
    sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="plain", choices=["plain", "surrogate", "rccl"])
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from meshvae_hip.engine import TrainStep  # noqa: E402  (NOT bench: importing it sets GPU_MAX_HW_QUEUES)

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if args.mode == "rccl":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from model import load_topology  # noqa: E402
from models.cheb_VAE import cheb_VAE  # noqa: E402
CFG_5K = {"n_layers": 4, "num_conv_filters": [16, 16, 16, 32, 32], "polygon_order": [6, 6, 6, 6, 6],
          "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}
D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
torch.manual_seed(666)
net = cheb_VAE(3, dict(CFG_5K), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
B = args.batch
step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=False, m_type="train", noise_seed=0,
                 rehearse_allreduce=args.mode == "rccl")
x = torch.randn(B, net.num_nodes[0], 3, generator=torch.Generator().manual_seed(0))
step.x.copy_(x)
step.x_gt = x.double().to(dev)
step.y.copy_(torch.nn.functional.one_hot(torch.arange(B) % 2, 2))

extra = torch.cuda.Stream(dev)
buf = torch.empty(48 << 20, dtype=torch.uint8, device=dev).view(torch.float32)      # ~30 us of elementwise work
ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()


def one_step():
    if args.mode != "surrogate":
        return step.step()
    step._draw_noise()
    step._fwd_bwd()
    main = torch.cuda.current_stream(dev)
    ev_a.record(main)
    extra.wait_event(ev_a)
    with torch.cuda.stream(extra):
        buf.mul_(1.0000001)
    ev_b.record(extra)
    main.wait_event(ev_b)
    step.opt.step(1.0)
    return step.out


for i in range(300):
    one_step()
    if i % 50 == 49:
        torch.cuda.synchronize(dev)
torch.cuda.synchronize(dev)
host = np.zeros(args.steps)
t0 = time.perf_counter()
for i in range(args.steps):
    h0 = time.perf_counter()
    one_step()
    host[i] = time.perf_counter() - h0
    if i % 100 == 99:                      # (bounded run-ahead: a stall must show in the step where it happens)
        torch.cuda.synchronize(dev)
torch.cuda.synchronize(dev)
dt = time.perf_counter() - t0
us = host * 1e6
print(f"mode {args.mode:9s} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')}: {1e3 * dt / args.steps:.4f} ms/step "
      f"({B * args.steps / dt:.0f} meshes/s); host enqueue per step: median {np.median(us):.0f} us, p99 {np.percentile(us, 99):.0f} us, "
      f"max {us.max():.0f} us, steps >= 1 ms: {int((us >= 1000).sum())} of {args.steps} (step index mod 16 of those: "
      f"{sorted(set(int(i) % 16 for i in np.flatnonzero(us >= 1000)))}); loss {float(step.out[0]):.4f}")
if dist.is_initialized():
    dist.destroy_process_group()
