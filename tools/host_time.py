#!/usr/bin/env python3
"""Host enqueue time vs wall time of one train step (is the step launch-bound?).

    python tools/host_time.py [--batch 64]
Times groups of 4 step() calls right after a synchronize (so nothing on the host waits for the GPU)
and compares with the steady-state wall time per step."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--micro", type=int, default=1)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    import bench
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    net = bench.build_model(dev)
    net.train()
    B = args.batch
    step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=args.graph, m_type="train", n_micro=args.micro)
    x = torch.randn(B, 4998, 3)
    step.x.copy_(x)
    step.x_gt = x.double().to(dev)
    step.y.copy_(torch.nn.functional.one_hot(torch.arange(B) % 2, 2))
    for _ in range(20):
        step.step()
    torch.cuda.synchronize()
    host = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step.step()
        host.append((time.perf_counter() - t0) / 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    host.sort()
    print(f"B={B} micro={args.micro} graph={args.graph}: host enqueue {1e6 * host[len(host) // 2]:.0f} us/step (min {1e6 * host[0]:.0f}), "
          f"wall {1e6 * wall:.0f} us/step")


if __name__ == "__main__":
    main()
