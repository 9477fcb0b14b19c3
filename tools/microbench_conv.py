#!/usr/bin/env python3
"""Micro-benchmark of one ChebConv layer through the C ABI (for rocprofv3 / PMC runs).

    python tools/microbench_conv.py --level 0 --cin 16 --cout 16 --batch 64 --iters 20 [--bwd]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topology", default="topology_5k.npz", help="hierarchy fixture under tests/golden")
    ap.add_argument("--level", type=int, default=0)
    ap.add_argument("--cin", type=int, default=16)
    ap.add_argument("--cout", type=int, default=16)
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--dwonly", action="store_true", help="backward without dX (weight/bias gradient kernels only)")
    ap.add_argument("--relu", type=int, default=1)
    ap.add_argument("--signs", type=int, default=1, help="fused-ReLU ops keep the ReLU signs as bytes (the train step's form)")
    args = ap.parse_args()

    from meshvae_hip import check, lib, topology
    from meshvae_hip.functional import workspace
    from nn.conv import ChebConv_batch
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(ROOT, "tests", "golden", args.topology))
    N = int(z["num_nodes"][args.level])
    ei = torch.from_numpy(np.vstack([z[f"A{args.level}_row"], z[f"A{args.level}_col"]]).astype(np.int64)).to(dev)
    ei, nrm = ChebConv_batch.norm(ei, N)
    op = topology.laplacian(ei, nrm, N)
    L = lib()
    B, Cin, Cout, K = args.batch, args.cin, args.cout, args.k
    x = torch.randn(B, N, Cin, device=dev)
    W = torch.randn(K, Cin, Cout, device=dev) * 0.1
    bias = torch.randn(Cout, device=dev) * 0.1
    out = torch.empty(B, N, Cout, device=dev)
    dout = torch.randn(B, N, Cout, device=dev)
    dx, dW, db = torch.empty_like(x), torch.empty_like(W), torch.empty_like(bias)
    wsb = max(L.mvh_cheb_conv_ws_bytes(B, N, Cin, Cout, K), L.mvh_cheb_conv_bwd_ws_bytes(B, N, Cin, Cout, K))
    ws = workspace(wsb, dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    use_signs = bool(args.signs and args.relu and Cout % 4 == 0 and K > 1)
    signs = torch.empty(B, N, max(Cout // 4, 1), dtype=torch.uint8, device=dev)

    def fwd():
        if use_signs:
            check(L.mvh_cheb_conv_fwd_signs(st, op.fwd.ref, x.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                            signs.data_ptr(), B, N, Cin, Cout, K, ws.data_ptr(), wsb))
        else:
            check(L.mvh_cheb_conv_fwd(st, op.fwd.ref, x.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), None,
                                      B, N, Cin, Cout, K, args.relu, ws.data_ptr(), wsb))

    def bwd():
        dxp = None if args.dwonly else dx.data_ptr()
        if use_signs:
            check(L.mvh_cheb_conv_bwd_signs(st, op.fwd.ref, op.bwd.ref, x.data_ptr(), W.data_ptr(), out.data_ptr(),
                                            signs.data_ptr(), dout.data_ptr(), dxp, dW.data_ptr(), db.data_ptr(),
                                            B, N, Cin, Cout, K, ws.data_ptr(), wsb))
        else:
            check(L.mvh_cheb_conv_bwd(st, op.fwd.ref, op.bwd.ref, x.data_ptr(), W.data_ptr(), out.data_ptr(),
                                      dout.data_ptr(), None, dxp, dW.data_ptr(), db.data_ptr(),
                                      B, N, Cin, Cout, K, args.relu, ws.data_ptr(), wsb))

    fn = bwd if (args.bwd or args.dwonly) else fwd
    fwd()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    print(f"level {args.level} N={N} {Cin}->{Cout} K={K} B={B} {'dW' if args.dwonly else ('bwd' if args.bwd else 'fwd')}: {us:.1f} us/call")


if __name__ == "__main__":
    main()
