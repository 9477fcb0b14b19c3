#!/usr/bin/env python3
"""hipGraph capture of the MODULE path (TrainStep(native=False)): the fused single-node forward of cheb_VAE and the
per-module autograd path must both capture and give the same loss."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "mesh-vae_amd"))
sys.path.insert(0, os.path.join(_ROOT, "tests"))
from conftest import TINY_CFG, ROOT
from model import load_topology
from models.cheb_VAE import cheb_VAE
from meshvae_hip.engine import TrainStep
dev = torch.device("cuda:0")
D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), dev)
res = {}
for fused in (True, False):
    torch.manual_seed(666)
    net = cheb_VAE(3, dict(TINY_CFG, dropout=0.0), D, U, A, nn_).to(dev).train()
    net.fused_step = fused
    st = TrainStep(net, 4, use_graph=True, native=False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 162, 3, generator=g)
    st.load(x, x, torch.nn.functional.one_hot(torch.arange(4) % 2, 2))
    torch.manual_seed(5)
    for _ in range(3):
        out = st.step()
    torch.cuda.synchronize()
    res[fused] = float(out[0])
    print("fused" if fused else "per-module", "graph-captured module path loss:", res[fused])
assert abs(res[True] - res[False]) < 1e-3 * abs(res[False]), res
print("capture ok")
