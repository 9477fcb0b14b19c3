#!/usr/bin/env python3
"""What a reference-style train loop gets (main.py:60-96: net(data, x_gt, y) -> loss.backward() -> torch.optim.Adam):
the per-module autograd path against the fused single-node forward of cheb_VAE, 5k model.
    python tools/module_path_step.py [--batch 64]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=100)
    a = ap.parse_args()
    from conftest import CFG_5K
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
    B = a.batch
    x = torch.randn(B, nn_[0], 3).to(dev)
    x_gt = x.double()
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)

    class Data:
        pass
    d = Data()
    d.x, d.num_graphs, d.edge_index = x.reshape(-1, 3), B, None
    for fused in (False, True):
        torch.manual_seed(666)
        net = cheb_VAE(3, dict(CFG_5K), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
        net.fused_step = fused
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)

        def step():
            opt.zero_grad()
            loss = net(d, x_gt, y, m_type="train")[0]
            loss.backward()
            opt.step()
            return loss
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        print(f"reference-style loop, {'fused single-node forward' if fused else 'per-module autograd path  '}: "
              f"B={B} {ms:.3f} ms/step {B / ms * 1e3:.0f} meshes/s loss={float(loss.detach()):.1f}")


if __name__ == "__main__":
    main()
