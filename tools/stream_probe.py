#!/usr/bin/env python3
"""The native step's forward + backward enqueued by this thread on the null stream against a torch side stream
(tools/async_probe.py found the second 3 x slower): one figure each, under the MESHVAE_DEBUG / runtime environment
of the caller."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import bench  # noqa: E402
from meshvae_hip.engine import NativeStep  # noqa: E402

dev = torch.device("cuda:0")
B = 64
net = bench.build_model(dev).train()
nat = NativeStep(net, B, grads="external")
x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(0)).to(dev)
x_gt = x.double()
y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).float().to(dev)
eps = torch.randn(B, 16, device=dev)
u = torch.rand(B * nat.u_cols, device=dev)
f32 = dict(dtype=torch.float32, device=dev)
outs = (torch.empty((), dtype=torch.float64, device=dev), torch.empty((), dtype=torch.int64, device=dev),
        torch.empty(B, 4998, 3, **f32), torch.empty(B, **f32), torch.empty(B, dtype=torch.float64, device=dev),
        torch.empty(B, 16, **f32), torch.empty(B, 2, **f32), torch.empty(B, 16, **f32), torch.empty(B, 16, **f32))
params = [p for p in net.parameters()]
grads = [torch.zeros_like(p) for p in params]
G = (ctypes.c_void_p * len(params))(*[g.data_ptr() for g in grads])
which = sys.argv[1] if len(sys.argv) > 1 else "both"


def run(n):
    for _ in range(n):
        nat.run_forward(x, x_gt, y, eps, u, outs, None)
        nat.run_backward(x, x_gt, y, eps, u, None, outs[2], outs[6], outs[7], outs[8], G, None)


def timed(label, fn, n=300):
    fn(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / n * 1e6:.1f} us/step (host {th / n * 1e6:.0f})", flush=True)


if which in ("null", "both"):
    timed("null stream", run)
if which in ("side", "both"):
    side = torch.cuda.Stream(dev)

    def f(n):
        with torch.cuda.stream(side):
            run(n)
    timed("side stream", f)
if which == "both":
    timed("null stream", run)

if which == "thread":
    # forward on this thread, backward on ONE persistent other thread (what the autograd engine does), null stream
    import queue
    import threading
    qi, qo = queue.Queue(), queue.Queue()

    def worker():
        while True:
            item = qi.get()
            if item is None:
                return
            nat.run_backward(x, x_gt, y, eps, u, None, outs[2], outs[6], outs[7], outs[8], G, None)
            qo.put(1)
    th_ = threading.Thread(target=worker, daemon=True)
    th_.start()

    def split(n):
        for _ in range(n):
            nat.run_forward(x, x_gt, y, eps, u, outs, None)
            qi.put(1)
            qo.get()
    timed("fwd here, bwd on a persistent thread", split)
    timed("all here                            ", run)
    timed("fwd here, bwd on a persistent thread", split)
    qi.put(None)
