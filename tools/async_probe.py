#!/usr/bin/env python3
"""GPU-side cost of the asynchronous launcher: the SAME forward + backward launch sequences (NativeStep.run_forward /
run_backward, fixed buffers, no optimizer, no Python between the calls) enqueued by the calling thread on its stream
against the launcher's worker thread on its high-priority stream.  Wall time per step, 300 steps, alternating."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import bench  # noqa: E402
import meshvae_hip  # noqa: E402
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
lch = meshvae_hip.launcher(0)


def run(launcher, n):
    for _ in range(n):
        nat.run_forward(x, x_gt, y, eps, u, outs, launcher)
        nat.run_backward(x, x_gt, y, eps, u, None, outs[2], outs[6], outs[7], outs[8], G, launcher)


for mode, l in (("sync", None), ("async", lch)) * 3:
    if mode == "async" and l is None:
        continue
    run(l, 300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(l, 300)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:5s}: {dt / 300 * 1e6:.1f} us/step wall, caller's enqueue {th / 300 * 1e6:.1f} us/step, loss {float(outs[0]):.2f}", flush=True)

# ---- where the launcher's time goes: the same sequences (1) on a torch side stream from this thread, (2) on the null stream
# from a Python worker thread, (3) on a high-priority side stream from this thread
import threading  # noqa: E402


def timed(label, fn, n=300):
    fn(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / n * 1e6:.1f} us/step", flush=True)


side = torch.cuda.Stream(dev)
hp = torch.cuda.Stream(dev, priority=-1)


def on_stream(s):
    def f(n):
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            run(None, n)
        torch.cuda.current_stream(dev).wait_stream(s)
    return f


def in_thread(n):
    t = threading.Thread(target=lambda: run(None, n))
    t.start()
    t.join()


def in_thread_on(s):
    def f(n):
        def body():
            with torch.cuda.stream(s):
                run(None, n)
        t = threading.Thread(target=body)
        t.start()
        t.join()
        torch.cuda.current_stream(dev).wait_stream(s)
    return f


timed("sync, this thread, null stream        ", lambda n: run(None, n))
timed("sync, this thread, side stream        ", on_stream(side))
timed("sync, this thread, high-prio stream   ", on_stream(hp))
timed("sync, other thread, null stream       ", in_thread)
timed("sync, other thread, side stream       ", in_thread_on(side))
timed("sync, other thread, high-prio stream  ", in_thread_on(hp))
