#!/usr/bin/env python3
"""Where the reference-API train loop (main.py:67-81,251) spends its time: host enqueue time of each phase
(zero_grad / net() / backward / optimizer.step, perf_counter without device syncs) beside the wall time per step,
for torch.optim.Adam in its default (foreach), fused and single-tensor forms, and engine.TrainStep for scale.
    python tools/ref_loop_probe.py [--batch 64] [--steps 200]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warm", type=int, default=300)
    a = ap.parse_args()
    import bench
    if os.environ.get("PROBE_ST_AUTOGRAD") == "1":        # backward on the calling thread (no hop to autograd's device thread)
        torch.autograd.set_multithreading_enabled(False)
    dev = torch.device("cuda:0")
    B = a.batch
    if "PROBE_LAUNCHER_LANES" in os.environ:               # 0: launcher jobs run their gradient work inline; 1: lowest-priority lanes
        from meshvae_hip import check, lib
        check(lib().mvh_debug_set(b"launcher_lanes", int(os.environ["PROBE_LAUNCHER_LANES"])))
    x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    x_gt = x.double()
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    d = bench.RefBatch(x) if hasattr(bench, "RefBatch") else type("D", (), dict(x=x.reshape(-1, 3), num_graphs=B,
                                                                                 edge_index=None))()
    for label, kw in (("foreach (default)", {}), ("foreach, grad_mode=assign", {"_assign": True}), ("fused=True", {"fused": True}),
                      ("fused=True, assign", {"fused": True, "_assign": True})):
        net = bench.build_model(dev).train()
        if kw.pop("_assign", False):
            net.grad_mode = "assign"
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4, **kw)
        ph = [0.0] * 4

        def step(acc=False):
            t0 = time.perf_counter()
            opt.zero_grad()
            t1 = time.perf_counter()
            loss = net(d, x_gt, y, m_type="train")[0]
            t2 = time.perf_counter()
            loss.backward()
            t3 = time.perf_counter()
            opt.step()
            t4 = time.perf_counter()
            if acc:
                for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                    ph[i] += v
            return loss
        for i in range(a.warm):
            step()
            if i % 50 == 49:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step(True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        us = [1e6 * v / a.steps for v in ph]
        print(f"reference loop, Adam {label:26s}: {ms:.4f} ms/step  host us: zero_grad {us[0]:.0f} net() {us[1]:.0f} "
              f"backward {us[2]:.0f} opt.step {us[3]:.0f} (sum {sum(us):.0f})  loss {float(loss):.1f}", flush=True)
    from meshvae_hip.engine import TrainStep
    net = bench.build_model(dev).train()
    st = TrainStep(net, B, noise_seed=666)
    st.x.copy_(x)
    st.x_gt = x_gt
    st.y.copy_(y)
    for i in range(a.warm):
        st.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(f"engine.TrainStep: {ms:.4f} ms/step (host enqueue {1e6 * th / a.steps:.0f} us/step)")


if __name__ == "__main__":
    main()
