#!/usr/bin/env python3
"""What one gradient all-reduce costs inside a stream of kernels (1-rank RCCL group on one GPU):
host time per call and the GPU-side time a dependent chain loses to it.
    python tools/allreduce_probe.py
"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29545")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
buf = torch.zeros(712_704, device=dev)
small = torch.zeros(20_000, device=dev)
a = torch.zeros(1 << 16, device=dev)


def chain(n_k, mode):
    for _ in range(n_k):
        a.add_(1.0)
    if mode == "one":
        dist.all_reduce(buf)
    elif mode == "two":
        dist.all_reduce(buf)
        dist.all_reduce(small)
    elif mode == "small":
        dist.all_reduce(small)
    a.add_(1.0)


for mode in ("none", "one", "small", "two"):
    for _ in range(20):
        chain(20, mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = 0.0
    for _ in range(200):
        h0 = time.perf_counter()
        chain(20, mode)
        th += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{mode:6s}: {dt * 1e6:8.1f} us per chain of 21 tiny kernels (+collectives), host enqueue {th / 200 * 1e6:7.1f} us")
h = []
for _ in range(200):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    dist.all_reduce(buf)
    h.append(time.perf_counter() - h0)
print(f"host time of one dist.all_reduce call (idle GPU): median {sorted(h)[100] * 1e6:.1f} us")
dist.destroy_process_group()
