"""Diagnostic (tooling): gradient error of the native step against the float64 oracle as a function of the batch size
and of dropout -- where does the 1e-5 .. 1e-4 of the B = 64 run come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_gpu_b64 as T
from conftest import CFG_5K

dev = torch.device("cuda:0")
keys = ["cheb.0.weight", "cheb.1.weight", "cheb.3.weight", "enc_lin.weight", "z_mean.weight", "dec_lin.weight", "dec_lin_2.weight",
        "cheb_dec.0.weight", "cheb_dec.1.weight", "cheb_dec.2.weight", "cheb_dec.2.bias", "cheb_dec.3.weight", "cheb_dec.4.weight"]
for B in [int(a) for a in sys.argv[1:]] or [4, 16, 64]:
    for pdrop in (0.0, 0.2):
        T.P_DROP = pdrop
        net = T._build(CFG_5K, "topology_5k.npz", dev, dropout=pdrop).train()
        x, y, eps, g = T._inputs(net, B)
        H, flat = net.num_hidden, net.dec_lin_2.out_features
        drop_u = torch.rand(B * (3 * H + flat), generator=g)
        _, got = T._native(net, B, x, y, eps, drop_u)
        if pdrop == 0.0:
            drop_u = torch.ones_like(drop_u)
        truth = T._oracle(CFG_5K, "topology_5k.npz", net, x, y, eps, drop_u, H, flat, dtype=torch.float64)
        row = []
        for k in keys:
            t = truth["grads"][k]
            row.append(f"{k.replace('weight', 'w').replace('bias', 'b')}={float((got['grads'][k].double() - t).norm() / t.norm()):.1e}")
        rec = float((got["recon"].double() - truth["recon"]).abs().max())
        print(f"B={B:3d} p={pdrop}: recon maxabs {rec:.1e} z {float((got['z'].double() - truth['z']).abs().max()):.1e} | " + " ".join(row), flush=True)
