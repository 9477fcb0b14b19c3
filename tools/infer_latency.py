#!/usr/bin/env python3
"""BASELINE configs[4]: the inference-only path of crecon.py:170-192 (`estimate_diff`): encoder ->
classifier -> z_mean -> decoder for the predicted class AND for the opposite class, no_grad, through
the reference-API module (nn.conv / nn.pool / cheb_VAE methods), eager and hipGraph-captured,
latency at batch 1 / 32 / 256 on the 5k template.

    python tools/infer_latency.py [--batches 1,32,256] [--iters 200]
"""
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
    ap.add_argument("--batches", default="1,32,256")
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    net = bench.build_model(dev)
    net.eval()

    def estimate_diff(x):
        h = net.encoder(x)
        y_hat = net.classifier(h)
        y = torch.nn.functional.one_hot(y_hat.argmax(-1), 2).to(torch.float32)
        mu = net.z_mean(torch.cat([y, h], -1))
        return net.sample(y, mu), net.sample(1.0 - y, mu), y_hat

    for B in [int(b) for b in args.batches.split(",")]:
        x = torch.randn(B, 4998, 3, device=dev)
        with torch.no_grad():
            for _ in range(5):
                ref = estimate_diff(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                estimate_diff(x)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / args.iters
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):        # warm the workspaces of the capture stream
                for _ in range(3):
                    estimate_diff(x)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = estimate_diff(x)
            g.replay()
            torch.cuda.synchronize()
            for a, b in zip(out, ref):
                assert torch.equal(a, b), "hipGraph replay differs from the eager result"
            t0 = time.perf_counter()
            for _ in range(args.iters):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / args.iters
        print(f"B={B:4d}: eager {1e3 * eager:.3f} ms ({B / eager:9.0f} meshes/s)   hipGraph {1e3 * graph:.3f} ms "
              f"({B / graph:9.0f} meshes/s)   graph == eager bitwise")


if __name__ == "__main__":
    main()
