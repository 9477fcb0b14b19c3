#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[3] (the hi-res template) by RUNNING THE REFERENCE ITSELF.

TEST INFRASTRUCTURE, build container only (see make_golden.py for the rules: only data is written,
never reference source; nothing under tests/ or bench.py imports this file).

The reference ships no 20k template, so one is synthesised the way SURVEY 8(d) pins it: 1 -> 4
midpoint subdivision of template/template5k.obj (19 992 vertices, 39 984 faces), hierarchy from the
reference's own generate_transform_matrices with factors 4,4,4,4,4, model = cheb_VAE with
n_layers 5, filters 16,16,16,32,32,32 and K = 10 everywhere.

    python oracle/make_golden_20k.py        # ~10-20 min (the reference's decimator is O(queue) per collapse)

Fixtures
  topology_20k.npz   A/D/U of the 6-level hierarchy (COO order as model.py:42-47 hands it over)
  model_20k.npz      eval vectors + train(dropout=0) loss and gradient norms/heads at B=2
                     (weights are NOT stored: they come from torch.manual_seed(666) in the
                     constructor's RNG order, pinned by sd_abs_sum and a few slices)
"""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the stand-ins and imports the reference modules)

CFG_20K = {"n_layers": 5, "num_conv_filters": [16, 16, 16, 32, 32, 32], "polygon_order": [10] * 6,
           "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}


def subdivide(v, f):
    v = [p for p in v]
    cache, nf = {}, []

    def mid(a, b):
        key = (min(a, b), max(a, b))
        if key not in cache:
            v.append(0.5 * (v[a] + v[b]))
            cache[key] = len(v) - 1
        return cache[key]

    for a, b, c in f:
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
    return np.stack(v), np.asarray(nf, dtype=np.int64)


def main():
    torch.set_num_threads(8)
    v, f = G.refshim.read_obj(os.path.join(G.REF, "template", "template5k.obj"))
    v20, f20 = subdivide(v, f)
    print("subdivided", v20.shape, f20.shape, flush=True)
    t0 = time.time()
    M, A, D, U = G.hierarchy(v20, f20, [4, 4, 4, 4, 4])
    print("hierarchy", [len(m.v) for m in M], f"{time.time() - t0:.0f} s", flush=True)
    topo = G.pack_topology(M, A, D, U)
    np.savez_compressed(os.path.join(G.OUT, "topology_20k.npz"), **topo)

    A_, D_, U_, nn_ = G.sparse_lists(topo)
    B = 2
    out = {}
    torch.manual_seed(666)
    net = G.cheb_VAE(3, dict(CFG_20K), D_, U_, A_, nn_, model="optimal_sigma_VAE")
    sd = net.state_dict()
    out["sd_keys"] = np.asarray(list(sd.keys()))
    out["sd_abs_sum"] = np.float64(sum(float(t.double().abs().sum()) for t in sd.values()))
    out["sd_head/cheb.0.weight"] = sd["cheb.0.weight"].reshape(-1)[:64].numpy().copy()
    out["sd_head/dec_lin_2.weight"] = sd["dec_lin_2.weight"].reshape(-1)[:64].numpy().copy()
    N0 = nn_[0]
    x = torch.randn(B, N0, 3, generator=torch.Generator().manual_seed(0))
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, num_classes=2)
    data = types.SimpleNamespace(x=x.reshape(B * N0, 3), edge_index=None, num_graphs=B)
    net.eval()
    with torch.no_grad():
        h = net.encoder(x)
        loss, correct, recon, (kld, rec, z_), y_hat = net(data, x.clone(), y, m_type="test")
    out["eval/h"], out["eval/y_hat"], out["eval/z"] = h.numpy(), y_hat.numpy(), z_.numpy()
    out["eval/kld"], out["eval/rec"], out["eval/loss"] = kld.numpy(), rec.numpy(), loss.numpy()
    out["eval/recon_sum"] = np.float64(recon.double().sum())
    out["eval/recon_abs_sum"] = np.float64(recon.double().abs().sum())
    out["eval/recon_head"] = recon[:, :512].numpy().copy()
    out["eval/recon_tail"] = recon[:, -512:].numpy().copy()
    print("eval loss", float(loss), "kld", kld.numpy(), flush=True)

    cfg0 = dict(CFG_20K)
    cfg0["dropout"] = 0.0
    torch.manual_seed(666)
    net0 = G.cheb_VAE(3, cfg0, D_, U_, A_, nn_, model="optimal_sigma_VAE")
    net0.train()
    torch.manual_seed(123)
    loss, correct, recon, (kld, rec, z_), y_hat = net0(data, x.double(), y, m_type="train")   # fp64 x_gt as main.py:69
    loss.backward()
    out["train/loss"], out["train/kld"], out["train/rec"] = loss.detach().numpy(), kld.detach().numpy(), rec.detach().numpy()
    names = []
    for k, p in net0.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        out[f"train/gnorm/{k}"] = np.float64(p.grad.double().norm())
        out[f"train/grad_head/{k}"] = p.grad.reshape(-1)[:1024].numpy().copy()
    out["train/grad_names"] = np.asarray(names)
    np.savez_compressed(os.path.join(G.OUT, "model_20k.npz"), **out)
    for fn in ("topology_20k.npz", "model_20k.npz"):
        print(fn, os.path.getsize(os.path.join(G.OUT, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
