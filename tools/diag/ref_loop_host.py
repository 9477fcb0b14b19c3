#!/usr/bin/env python3
"""Host time of the pieces of the module path inside the reference-style loop (perf_counter around each, no device syncs)."""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import bench  # noqa: E402
import models.cheb_VAE as M  # noqa: E402
from meshvae_hip import engine  # noqa: E402

acc = collections.defaultdict(float)


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0
    setattr(obj, name, w)


dev = torch.device("cuda:0")
B = 64
net = bench.build_model(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(0)).to(dev)
x_gt = x.double()
y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
d = bench.RefBatch(x)


def step():
    opt.zero_grad()
    loss = net(d, x_gt, y, m_type="train")[0]
    loss.backward()
    opt.step()


for _ in range(50):
    step()
wrap(net, "_fused_entry")
wrap(net, "_host_eps")
wrap(net, "_forward_fused")
wrap(engine.NativeStep, "run_forward")
wrap(engine.NativeStep, "run_backward")
wrap(engine.NativeStep, "refresh_param_pointers")
wrap(torch, "rand")
fwd0, bwd0 = M._FusedModelFn.forward, M._FusedModelFn.backward


def tf(*a, **k):
    t0 = time.perf_counter()
    try:
        return fwd0(*a, **k)
    finally:
        acc["Fn.forward"] += time.perf_counter() - t0


def tb(*a, **k):
    t0 = time.perf_counter()
    try:
        return bwd0(*a, **k)
    finally:
        acc["Fn.backward"] += time.perf_counter() - t0


M._FusedModelFn.forward = staticmethod(tf)
M._FusedModelFn.backward = staticmethod(tb)
for _ in range(100):
    step()
torch.cuda.synchronize()
acc.clear()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
dt = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"loop host {dt / n * 1e6:.0f} us/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:26s} {v / n * 1e6:7.1f} us/step")
