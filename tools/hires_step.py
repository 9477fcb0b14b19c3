#!/usr/bin/env python3
"""Train-step time of BASELINE configs[3] (19 992-vertex template, 6 levels, K = 10) on one GPU.
    python tools/hires_step.py [--batch 8] [--steps 30]
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    from conftest import CFG_20K
    from meshvae_hip.engine import TrainStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_20k.npz"), dev)
    torch.manual_seed(666)
    net = cheb_VAE(3, dict(CFG_20K), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
    step = TrainStep(net, a.batch, use_graph=False)
    x = torch.randn(a.batch, nn_[0], 3)
    step.load(x, x, torch.nn.functional.one_hot(torch.arange(a.batch) % 2, 2))
    for _ in range(5):
        step.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(f"20k-vertex K=10 train step: B={a.batch} {ms:.3f} ms/step {a.batch / ms * 1e3:.0f} meshes/s "
          f"({a.batch / ms * 1e3 * 31.2e6 / 8e12 * 100:.1f} % of the 31.2 MB/mesh HBM bound), loss={float(step.out[0]):.1f}")


if __name__ == "__main__":
    main()
