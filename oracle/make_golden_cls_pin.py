#!/usr/bin/env python3
"""Pin of the crecon classifier's convolution against REFERENCE-HELD code.

TEST INFRASTRUCTURE, build container only.  models/cheb_cls.py imports torch-geometric's ChebConv, which this
image lacks; `oracle.refshim.PygChebConv` restates its published 2.0.4 algorithm and generated cls_*.npz.  The
reference tree carries its OWN copy of that operator -- class ChebConv at /root/reference/nn/conv.py:390-521
(edge-list Laplacian with self loops, per-order `weight[k]`, the vendored MessagePassing.propagate of :242-331) --
so this script runs that in-tree class, unmodified, on the weights of a PygChebConv (weight[k] = lins.k.weight^T,
same bias) and stores inputs, its outputs and its autograd gradients.  tests/test_oracle_golden.py then asserts
that PygChebConv reproduces them: the stand-in's recurrence, message/aggregate and weight layout are pinned by
reference code (its Laplacian helpers get_laplacian / add_self_loops remain elementary refshim restatements of
torch_geometric.utils, as for every other fixture).

    python oracle/make_golden_cls_pin.py    -> tests/golden/cheb_pin.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import make_golden as mg  # noqa: E402  (installs refshim, puts the reference on sys.path)
from oracle import refshim  # noqa: E402

import nn.conv as ref_conv  # noqa: E402  (reference, in-tree ChebConv)

CASES = [  # tag, level, B, Cin, Cout, K, bias
    ("cls_in", 0, 3, 6, 16, 6, True),        # the classifier's first layer shape (6-channel diff input)
    ("mid", 1, 2, 16, 32, 6, True),
    ("k1", 2, 2, 8, 8, 1, True),
    ("k2_nobias", 1, 2, 3, 5, 2, False),
    ("k3", 0, 1, 5, 7, 3, True),
]


def main():
    topo = dict(np.load(os.path.join(mg.OUT, "topology_tiny.npz")))
    out = {"cases": np.asarray([c[0] for c in CASES])}
    for tag, lvl, B, Cin, Cout, K, bias in CASES:
        N = int(topo["num_nodes"][lvl])
        ei = torch.from_numpy(np.vstack([topo[f"A{lvl}_row"], topo[f"A{lvl}_col"]]).astype(np.int64))
        torch.manual_seed(100 + len(tag))
        shim = refshim.PygChebConv(Cin, Cout, K, bias=bias)
        if bias:
            shim.bias.data.normal_(0, 0.3)
        intree = ref_conv.ChebConv(Cin, Cout, K, bias=bias)          # /root/reference/nn/conv.py:390
        with torch.no_grad():
            for k in range(K):
                intree.weight[k].copy_(shim.lins[k].weight.t())
            if bias:
                intree.bias.copy_(shim.bias)
        g = torch.Generator().manual_seed(7 + lvl)
        x = torch.randn(B, N, Cin, generator=g)
        gy = torch.randn(B, N, Cout, generator=g)
        xr = x.clone().requires_grad_(True)
        y = torch.stack([intree(xr[b], ei) for b in range(B)])        # in-tree node_dim = 0: one mesh per call
        y.backward(gy)
        out[f"{tag}/meta"] = np.asarray([lvl, B, Cin, Cout, K, int(bias)], dtype=np.int64)
        out[f"{tag}/x"], out[f"{tag}/gy"], out[f"{tag}/y"] = x.numpy(), gy.numpy(), y.detach().numpy()
        out[f"{tag}/gx"] = xr.grad.numpy()
        out[f"{tag}/weight"] = intree.weight.detach().numpy()          # [K, Cin, Cout]
        out[f"{tag}/gweight"] = intree.weight.grad.numpy()
        if bias:
            out[f"{tag}/bias"], out[f"{tag}/gbias"] = intree.bias.detach().numpy(), intree.bias.grad.numpy()
        # the stand-in on the same numbers, for the log only (the test repeats this comparison)
        ys = shim(x, ei)
        print(f"{tag:10s} N={N:4d} {Cin}->{Cout} K={K}: max|in-tree - PygChebConv| = {(ys - y).abs().max().item():.2e}")
    path = os.path.join(mg.OUT, "cheb_pin.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
