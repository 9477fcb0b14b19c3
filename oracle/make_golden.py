#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

TEST INFRASTRUCTURE.  Imports the unmodified reference modules from
``/root/reference`` (with ``oracle.refshim`` standing in for the third-party
packages this image lacks), runs them on seeded inputs and writes small ``.npz``
fixtures to ``tests/golden/``.  Only data (inputs + expected outputs) is
written -- never reference source.  The reference tree does not exist on the GPU
box, so nothing under ``tests/`` / ``bench.py`` imports this file at run time.

    python oracle/make_golden.py            # regenerates every fixture (~1-2 min)

Fixtures
  topology_5k.npz    A/D/U hierarchy of template/template5k.obj as the reference
                     hands it to cheb_VAE (model.py:42-47, COO order preserved)
  topology_tiny.npz  same for a 162-vertex icosphere (factors 4,4)
  ops_tiny.npz       ChebConv_batch / SurfacePool forward + autograd grads
  model_tiny.npz     full tiny cheb_VAE: state_dict, eval + train(dropout=0) vectors
  model_5k.npz       full default.cfg cheb_VAE on the 5k template, B=4
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MESHVAE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
sys.path.insert(0, REF)

import mesh_operations  # noqa: E402  (reference)
from models.cheb_VAE import cheb_VAE  # noqa: E402  (reference)
from nn.conv import ChebConv_batch  # noqa: E402  (reference)
from nn.pool import SurfacePool  # noqa: E402  (reference)
from model import scipy_to_torch_sparse  # noqa: E402  (reference)
import logpdf as ref_logpdf  # noqa: E402  (reference)


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# --------------------------------------------------------------------------- meshes
def icosphere(subdiv):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
         (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
         (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    v = np.stack(v)
    # make it non-spherical so quadric costs are not all tied
    v = v * np.array([1.0, 0.8, 1.3]) + 0.05 * np.sin(3.0 * v[:, [1, 2, 0]])
    return v, np.asarray(f, dtype=np.int64)


def hierarchy(v, f, factors):
    mesh = refshim.Mesh(v=v, f=f)
    M, A, D, U = mesh_operations.generate_transform_matrices(mesh, factors)
    return M, A, D, U


def pack_topology(M, A, D, U):
    d = {"n_levels": np.int64(len(M)), "num_nodes": np.asarray([len(m.v) for m in M], dtype=np.int64)}
    for i, a in enumerate(A):
        t = scipy_to_torch_sparse(a)           # exactly what model.py:44-46 builds
        idx = t._indices().numpy()
        ei, nrm = ChebConv_batch.norm(t._indices(), len(M[i].v))
        assert np.array_equal(ei.numpy(), idx)  # no self loops in any level
        d[f"A{i}_row"], d[f"A{i}_col"] = idx[0].astype(np.int32), idx[1].astype(np.int32)
        d[f"A{i}_val"] = t._values().numpy()
        d[f"A{i}_norm"] = nrm.numpy()
    for name, mats in (("D", D), ("U", U)):
        for i, m in enumerate(mats):
            t = scipy_to_torch_sparse(m)
            idx = t._indices().numpy()
            d[f"{name}{i}_row"], d[f"{name}{i}_col"] = idx[0].astype(np.int32), idx[1].astype(np.int32)
            d[f"{name}{i}_val"] = t._values().numpy()
            d[f"{name}{i}_shape"] = np.asarray(m.shape, dtype=np.int64)
    return d


def sparse_lists(topo):
    n = int(topo["n_levels"])
    nn_ = [int(x) for x in topo["num_nodes"]]

    def coo(name, i, shape):
        idx = torch.from_numpy(np.vstack([topo[f"{name}{i}_row"], topo[f"{name}{i}_col"]]).astype(np.int64))
        return torch.sparse_coo_tensor(idx, torch.from_numpy(topo[f"{name}{i}_val"]), shape,
                                       check_invariants=False)

    A = [coo("A", i, (nn_[i], nn_[i])) for i in range(n)]
    D = [coo("D", i, (nn_[i + 1], nn_[i])) for i in range(n - 1)]
    U = [coo("U", i, (nn_[i], nn_[i + 1])) for i in range(n - 1)]
    return A, D, U, nn_


# --------------------------------------------------------------------------- per-op vectors
def conv_case(tag, topo, level, N_x, B, Cin, Cout, K, bias, seed, out):
    g = torch.Generator().manual_seed(seed)
    ei = torch.from_numpy(np.vstack([topo[f"A{level}_row"], topo[f"A{level}_col"]]).astype(np.int64))
    ei, nrm = ChebConv_batch.norm(ei, int(topo["num_nodes"][level]))
    conv = ChebConv_batch(Cin, Cout, K, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(K, Cin, Cout, generator=g) * 0.3)
        if bias:
            conv.bias.copy_(torch.randn(Cout, generator=g) * 0.3)
    x = torch.randn(B, N_x, Cin, generator=g, requires_grad=True)
    y = conv(x, ei, nrm)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out[f"{tag}_meta"] = np.asarray([level, N_x, B, Cin, Cout, K, int(bias)], dtype=np.int64)
    out[f"{tag}_x"] = x.detach().numpy()
    out[f"{tag}_w"] = conv.weight.detach().numpy()
    if bias:
        out[f"{tag}_b"] = conv.bias.detach().numpy()
        out[f"{tag}_gb"] = conv.bias.grad.numpy()
    out[f"{tag}_y"] = y.detach().numpy()
    out[f"{tag}_gy"] = gy.numpy()
    out[f"{tag}_gx"] = x.grad.numpy()
    out[f"{tag}_gw"] = conv.weight.grad.numpy()


def pool_case(tag, mat, B, C, seed, out):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, mat.shape[1], C, generator=g, requires_grad=True)
    y = SurfacePool()(x, mat)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out[f"{tag}_x"], out[f"{tag}_y"] = x.detach().numpy(), y.detach().numpy()
    out[f"{tag}_gy"], out[f"{tag}_gx"] = gy.numpy(), x.grad.numpy()


# --------------------------------------------------------------------------- full model vectors
def model_vectors(topo, config, B, store_full, out):
    A, D, U, nn_ = sparse_lists(topo)
    torch.manual_seed(666)
    net = cheb_VAE(3, dict(config), D, U, A, nn_, model="optimal_sigma_VAE")
    sd = net.state_dict()
    out["sd_keys"] = np.asarray(list(sd.keys()))
    for k, v in sd.items():
        out[f"sd/{k}"] = v.numpy().copy()
    out["sd_abs_sum"] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))

    N0 = nn_[0]
    x = torch.randn(B, N0, 3, generator=torch.Generator().manual_seed(0))
    labels = torch.arange(B) % 2
    y = torch.nn.functional.one_hot(labels, num_classes=2)       # int64, as main.py:71
    data = types.SimpleNamespace(x=x.reshape(B * N0, 3), edge_index=None, num_graphs=B)
    out["x"], out["y"] = x.numpy(), y.numpy()

    # ---- eval mode (main.py:129): float32 x_gt (inference.py:87) and float64 x_gt (main.py:69)
    net.eval()
    with torch.no_grad():
        h = net.encoder(x)
        cheb0 = torch.relu(net.cheb[0](x, net.A_edge_index[0], net.A_norm[0]))
        loss, correct, recon, (kld, rec, z_), y_hat = net(data, x.clone(), y, m_type="test")
        loss64, _, _, (_, rec64, _), _ = net(data, x.double(), y, m_type="test")
        oppo = net.sample(1 - y, z_)
        mu = net.z_mean(torch.cat([y, h], -1))
        logvar = net.z_log_var(torch.cat([y, h], -1))
    out["eval/cheb0_sum"] = np.float64(cheb0.double().sum())
    out["eval/cheb0_abs_sum"] = np.float64(cheb0.double().abs().sum())
    out["eval/cheb0_slice"] = cheb0[:, :64].numpy()
    out["eval/h"], out["eval/y_hat"] = h.numpy(), y_hat.numpy()
    out["eval/mu"], out["eval/logvar"] = mu.numpy(), logvar.numpy()
    out["eval/kld"], out["eval/rec"] = kld.numpy(), rec.numpy()
    out["eval/loss"], out["eval/correct"] = loss.numpy(), correct.numpy()
    out["eval/loss64"], out["eval/rec64"] = loss64.numpy(), rec64.numpy()
    out["eval/recon"], out["eval/z"] = recon.numpy(), z_.numpy()
    out["eval/oppo_recon"] = oppo.numpy()

    # ---- train mode, dropout p=0 so the only randomness is the host-side eps (cheb_VAE.py:316)
    cfg0 = dict(config)
    cfg0["dropout"] = 0.0
    torch.manual_seed(666)
    net0 = cheb_VAE(3, cfg0, D, U, A, nn_, model="optimal_sigma_VAE")
    net0.train()
    torch.manual_seed(123)
    eps_expected = torch.normal(mean=0, std=1, size=(B, config["num_style"]))
    torch.manual_seed(123)
    loss, correct, recon, (kld, rec, z_), y_hat = net0(data, x.clone(), y, m_type="train")
    loss.backward()
    out["train/eps"] = eps_expected.numpy()
    out["train/loss"], out["train/kld"], out["train/rec"] = loss.detach().numpy(), kld.detach().numpy(), rec.detach().numpy()
    out["train/z"], out["train/y_hat"] = z_.detach().numpy(), y_hat.detach().numpy()
    out["train/recon_sum"] = np.float64(recon.detach().double().sum())
    out["train/recon_slice"] = recon.detach()[:, :64].numpy()
    gnames = []
    for k, p in net0.named_parameters():
        if p.grad is None:
            continue
        gnames.append(k)
        out[f"train/gnorm/{k}"] = np.float64(p.grad.double().norm())
        if store_full or p.grad.numel() <= 4096:
            out[f"train/grad/{k}"] = p.grad.numpy().copy()
        else:
            out[f"train/grad_head/{k}"] = p.grad.reshape(-1)[:4096].numpy().copy()
    out["train/grad_names"] = np.asarray(gnames)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    # ---------------- 5k template
    v, f = refshim.read_obj(os.path.join(REF, "template", "template5k.obj"))
    print("template", v.shape, f.shape)
    M, A, D, U = hierarchy(v, f, [4, 4, 4, 4])
    topo5k = pack_topology(M, A, D, U)
    for i in range(5):
        h = sha16(np.vstack([topo5k[f"A{i}_row"], topo5k[f"A{i}_col"]]).astype(np.int64))
        print(f"A{i} N={topo5k['num_nodes'][i]} E={len(topo5k[f'A{i}_row'])} sha={h}")
    np.savez_compressed(os.path.join(OUT, "topology_5k.npz"), **topo5k)

    # ---------------- tiny icosphere
    tv, tf = icosphere(2)
    M, A, D, U = hierarchy(tv, tf, [4, 4])
    topot = pack_topology(M, A, D, U)
    topot["verts"], topot["faces"] = tv, tf
    print("tiny levels", topot["num_nodes"])
    np.savez_compressed(os.path.join(OUT, "topology_tiny.npz"), **topot)

    # ---------------- per-op vectors on the tiny hierarchy
    ops = {}
    N0, N1 = int(topot["num_nodes"][0]), int(topot["num_nodes"][1])
    conv_case("c_3_16_k6", topot, 0, N0, 3, 3, 16, 6, True, 1, ops)
    conv_case("c_16_16_k6", topot, 0, N0, 2, 16, 16, 6, True, 2, ops)
    conv_case("c_16_32_k6", topot, 1, N1, 3, 16, 32, 6, True, 3, ops)
    conv_case("c_32_16_k6", topot, 1, N1, 2, 32, 16, 6, True, 4, ops)
    conv_case("c_32_32_k10", topot, 0, N0, 2, 32, 32, 10, True, 5, ops)
    conv_case("c_5_7_k1", topot, 0, N0, 2, 5, 7, 1, True, 6, ops)
    conv_case("c_5_7_k2", topot, 0, N0, 2, 5, 7, 2, True, 7, ops)
    conv_case("c_16_3_k6_nobias", topot, 0, N0, 2, 16, 3, 6, False, 8, ops)
    # the final-layer quirk (cheb_VAE.py:288): coarsest-level edges on the finest tensor
    conv_case("c_quirk_16_3_k6", topot, 2, N0, 2, 16, 3, 6, False, 9, ops)
    ops["case_names"] = np.asarray(sorted({k.rsplit("_", 1)[0] for k in ops if k.endswith("_meta")}))
    At, Dt, Ut, _ = sparse_lists(topot)
    pool_case("p_D0", Dt[0], 3, 16, 11, ops)
    pool_case("p_D1", Dt[1], 2, 32, 12, ops)
    pool_case("p_U0", Ut[0], 3, 16, 13, ops)
    pool_case("p_U1", Ut[1], 2, 32, 14, ops)
    # logpdf scalars (logpdf.py:7-8,22-28)
    g = torch.Generator().manual_seed(21)
    mu, lv = torch.randn(5, 16, generator=g), torch.randn(5, 16, generator=g)
    ops["kld_mu"], ops["kld_lv"], ops["kld_out"] = mu.numpy(), lv.numpy(), ref_logpdf.KLD(mu, lv).numpy()
    ls = ref_logpdf.softclip(torch.Tensor([1]), -6)
    ops["log_sigma"] = ls.numpy()
    a, b = torch.randn(2, 7, 3, generator=g), torch.randn(2, 7, 3, generator=g)
    ops["nll_mu"], ops["nll_x"], ops["nll_out"] = a.numpy(), b.numpy(), ref_logpdf.gaussian_nll(a, ls, b).numpy()
    np.savez_compressed(os.path.join(OUT, "ops_tiny.npz"), **ops)

    # ---------------- full models
    tiny_cfg = {"n_layers": 2, "num_conv_filters": [8, 16, 16], "polygon_order": [6, 6, 6],
                "num_classes": 2, "num_style": 16, "num_hidden": 64, "dropout": 0.2}
    mt = {}
    model_vectors(topot, tiny_cfg, 4, True, mt)
    np.savez_compressed(os.path.join(OUT, "model_tiny.npz"), **mt)

    cfg5k = {"n_layers": 4, "num_conv_filters": [16, 16, 16, 32, 32], "polygon_order": [6, 6, 6, 6, 6],
             "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}
    m5 = {}
    model_vectors(topo5k, cfg5k, 4, True, m5)
    print("5k: sum|params| =", m5["sd_abs_sum"], " cheb0 sum =", m5["eval/cheb0_sum"],
          " y_hat[0] =", m5["eval/y_hat"][0], " kld =", m5["eval/kld"], "loss", m5["eval/loss"])
    np.savez_compressed(os.path.join(OUT, "model_5k.npz"), **m5)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
