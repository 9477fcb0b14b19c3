#!/usr/bin/env python3
"""Does the HIP runtime scale kernel launches over host threads?  T threads, each on its own
stream, enqueue tiny kernels through the C ABI (mvh_adam_step on 64 floats = 2 launches)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from meshvae_hip import lib  # noqa: E402


def worker(n, out, i, barrier):
    dev = torch.device("cuda:0")
    with torch.cuda.device(dev):
        st = torch.cuda.Stream(dev)
        p, g, m, v = (torch.zeros(64, device=dev) for _ in range(4))
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        L = lib()
        args = (st.cuda_stream, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 64, 1e-3, 0.9, 0.999, 1e-8, 0.0,
                1.0, cnt.data_ptr())
        L.mvh_adam_step(*args)
        torch.cuda.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(n):
            L.mvh_adam_step(*args)
        out[i] = time.perf_counter() - t0
        st.synchronize()


def main():
    lib()
    torch.zeros(1, device="cuda:0")
    n = 5000
    for T in (1, 2, 4):
        out = [0.0] * T
        barrier = threading.Barrier(T)
        th = [threading.Thread(target=worker, args=(n, out, i, barrier)) for i in range(T)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        print(f"T={T}: per-thread {1e6 * max(out) / (2 * n):.2f} us/launch, aggregate {2 * n * T / max(out) / 1e3:.0f} k launches/s")


if __name__ == "__main__":
    main()
