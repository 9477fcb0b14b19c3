#!/usr/bin/env python3
"""The train step with its batch handed over as HOST buffers every step (what a DataLoader-fed loop does): pinned x fp32
[B,4998,3], x_gt fp64, one-hot y copied on the step's stream (TrainStep.load) -- never `value` of the bench line, noted in
DESIGN.md section 8 as the PCIe-inclusive rate.  Also with the copies on a second stream, one step ahead (double buffer)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import bench  # noqa: E402
from meshvae_hip.engine import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
B = 64
net = bench.build_model(dev).train()
step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, noise_seed=666)
step.x_gt = torch.zeros(B, 4998, 3, dtype=torch.float64, device=dev)
hx = [torch.randn(B, 4998, 3).pin_memory() for _ in range(2)]
hg = [h.double().pin_memory() for h in hx]
hy = [torch.nn.functional.one_hot(torch.arange(B) % 2, 2).float().pin_memory() for _ in range(2)]


def run(mode, n):
    copy = torch.cuda.Stream(dev)
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    for i in range(n):
        k = i & 1
        if mode == "resident":
            pass
        elif mode == "same stream":
            step.load(hx[k], hg[k], hy[k])
        else:                                  # copy stream, overlapping the previous step's compute (one staging set per parity)
            with torch.cuda.stream(copy):
                copy.wait_stream(torch.cuda.current_stream(dev)) if i < 2 else None
                stage[k][0].copy_(hx[k], non_blocking=True)
                stage[k][1].copy_(hg[k], non_blocking=True)
                stage[k][2].copy_(hy[k], non_blocking=True)
                ev[k].record(copy)
            torch.cuda.current_stream(dev).wait_event(ev[k])
            step.x, step.x_gt, step.y = stage[k]
        step.step()


stage = [(torch.zeros(B, 4998, 3, device=dev), torch.zeros(B, 4998, 3, dtype=torch.float64, device=dev),
          torch.zeros(B, 2, device=dev)) for _ in range(2)]
for mode in ("resident", "same stream", "copy stream", "resident"):
    run(mode, 300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(mode, 500)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 500 * 1e3
    print(f"{mode:12s}: {ms:.4f} ms/step  {B / ms * 1e3:.0f} meshes/s  ({11.5 / ms:.1f} GB/s of host buffers)" if mode != "resident"
          else f"{mode:12s}: {ms:.4f} ms/step  {B / ms * 1e3:.0f} meshes/s", flush=True)
