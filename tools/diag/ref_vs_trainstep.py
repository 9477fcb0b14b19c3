#!/usr/bin/env python3
"""Per-tensor distance between the reference-style loop (module path + torch.optim.Adam) and engine.TrainStep after 1..3
steps on the 5k model (same host noise, dropout 0): gradients of step 1 and parameters after every step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import bench  # noqa: E402
from meshvae_hip.engine import TrainStep, _Batch  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(1)).to(dev)
x_gt = x.double()
y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)


def model():
    net = bench.build_model(dev).train()
    net.dropout.p = 0.0
    return net


net = model()
step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=False)
step.load(x, x, y)
step.x_gt = x_gt
torch.manual_seed(5)
A = []
for i in range(3):
    step.step()
    if i == 0:
        gA = step.flat.grad.clone()
        offs = step.flat.offsets
    A.append({k: v.detach().clone() for k, v in net.state_dict().items()})
net2 = model()
opt = torch.optim.Adam(net2.parameters(), lr=1e-3, weight_decay=5e-4)
torch.manual_seed(5)
for i in range(3):
    opt.zero_grad()
    loss = net2(_Batch(x), x_gt, y, m_type="train")[0]
    loss.backward()
    if i == 0:
        for (k, p), o in zip(net2.named_parameters(), offs):
            if p.grad is not None:
                ga = gA[o:o + p.numel()].view_as(p)
                d = (ga - p.grad).abs().max().item()
                print(f"grad step1 {k:28s} max|d| {d:.3e}  max|g| {p.grad.abs().max().item():.3e}  min|g| {p.grad.abs().min().item():.3e}")
    opt.step()
    for k, v in net2.state_dict().items():
        d = (v - A[i][k]).abs()
        if d.max().item() > 1e-7:
            j = int(d.argmax())
            print(f"step {i + 1} {k:28s} max|dp| {d.max().item():.3e} at {j}  n(>1e-6) {(d > 1e-6).sum().item()}")
