#!/usr/bin/env python3
"""Time one crecon classifier training step (crecon.py:65-100) on the HIP path: estimate_diff of the frozen
VAE (encode + classify + two decodes, no grad) -> cheb_GCN forward -> cross-entropy -> backward -> Adam.
    python tools/crecon_step.py [--batch 16] [--steps 50]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--engine", choices=["module", "eager", "graph"], default="module",
                    help="module: torch.optim.Adam over the autograd modules; eager/graph: engine.ClassifierStep")
    a = ap.parse_args()
    from crecon_ops import estimate_diff
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    from models.cheb_cls import cheb_GCN
    dev = torch.device("cuda:0")
    cfg = {"n_layers": 4, "num_conv_filters": [16, 16, 16, 32, 32], "polygon_order": [6] * 5,
           "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
    torch.manual_seed(666)
    vae = cheb_VAE(3, dict(cfg), D, U, A, nn_, model="optimal_sigma_VAE").to(dev)
    net = cheb_GCN(6, dict(cfg, num_conv_filters=list(cfg["num_conv_filters"])), D, U, A, nn_).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=5e-4)
    crit = torch.nn.CrossEntropyLoss()
    x = torch.randn(a.batch, nn_[0], 3, device=dev)
    y = (torch.arange(a.batch, device=dev) % 2)

    if a.engine != "module":
        from meshvae_hip.engine import ClassifierStep
        cs = ClassifierStep(net, vae, a.batch, use_graph=(a.engine == "graph"))
        cs.load(x, y)

    def step():
        if a.engine != "module":
            return cs.step()[0]
        diff, _ = estimate_diff(vae, x, y, "train")
        opt.zero_grad()
        loss = crit(net(diff), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(f"crecon classifier step [{a.engine}]: B={a.batch} {ms:.3f} ms/step  {a.batch / ms * 1e3:.0f} meshes/s  loss={float(loss.detach()):.4f}")


if __name__ == "__main__":
    main()
