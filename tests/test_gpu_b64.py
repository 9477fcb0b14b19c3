"""GPU (-m gpu): oracle parity AT THE BENCHMARKED BATCH SIZE (B = 64) -- every mesh of the batch, every path the
bench numbers come from (VERDICT r2 "what's weak" #1).

The other parity files run at B <= 8; the block -> mesh maps of the chip-filling kernels (mesh = (jj / NS) * 8 + xcd in
cheb_lds / cheb_l0h / cheb_big, k_spmm's mesh -> XCD tiling, k_reduce_partials' 1024-thread form at G >= 128) only take
their large-batch branches beyond that.  Here:
  * fp32 native step, 5k template, B = 64, dropout 0.2 under shared masks: recon / z / y_hat of ALL 64 meshes, the loss
    and every gradient against the CPU oracle at the 1e-4 bars of north_star;
  * bf16-storage step against the fp32 HIP step at B = 64 (the bf16 bars of test_gpu_bf16.py);
  * 20k template (BASELINE configs[3]), K = 10, B = 64 against the oracle (meshes 8..63 of cheb_big, G >= 128);
  * B = 64 == the concatenation of eight B = 8 runs, BITWISE, for the level-0 conv forward / dX and for every per-mesh
    output of the whole step (a mesh permutation or a dropped mesh cannot survive it).
"""
import os

import numpy as np
import pytest
import torch

from conftest import CFG_5K, CFG_20K, ROOT, grad_bar

pytestmark = pytest.mark.gpu
FWD_ATOL = 1e-4
P_DROP = 0.2


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _build(cfg, topo_name, dev, dropout=P_DROP):
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", topo_name), dev)
    torch.manual_seed(666)
    return cheb_VAE(3, dict(cfg, dropout=dropout), D, U, A, nn_, model="optimal_sigma_VAE").to(dev)


def _inputs(net, B, seed=31):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, net.num_nodes[0], 3, generator=g)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2)
    eps = torch.randn(B, net.z, generator=g)
    return x, y, eps, g


def _drop_blocks(drop_u, B, H, flat):
    """The library's layout of the dropout uniforms: four contiguous blocks [B, H] (encoder head) | [B, H]
    (classifier) | [B, H] (dec_lin) | [B, flat] (dec_lin_2)."""
    return [drop_u[0:B * H].reshape(B, H), drop_u[B * H:2 * B * H].reshape(B, H),
            drop_u[2 * B * H:3 * B * H].reshape(B, H), drop_u[3 * B * H:].reshape(B, flat)]


def _native(net, B, x, y, eps, drop_u, storage="f32", gt64=False):
    from meshvae_hip.engine import NativeStep
    dev = next(net.parameters()).device
    for p in net.parameters():
        p.grad = None
    nat = NativeStep(net, B, storage=storage)
    xd = x.to(dev)
    loss, corr, recon, (kld, rec, z_), yh = nat.forward_backward(xd, xd.double() if gt64 else xd, y.to(dev),
                                                                 eps=eps.to(dev), drop_u=drop_u.to(dev))
    torch.cuda.synchronize()
    out = dict(loss=float(loss), correct=int(corr), recon=recon.cpu().clone(), kld=kld.cpu().clone(), rec=rec.cpu().clone(),
               z=z_.cpu().clone(), y_hat=yh.cpu().clone(),
               grads={k: q.grad.cpu().clone() for k, q in net.named_parameters() if q.grad is not None})
    return nat, out


def _oracle(cfg, topo_name, net, x, y, eps, drop_u, H, flat, dtype=torch.float32, pins=None):
    """dtype = float32: the reference's arithmetic (the parity bar).  float64: the exact answer, used only to print how
    far this library and the reference's own fp32 rounding each are from it.
    Records the pre-activation of every ReLU site (result["pre"][site]); pins = {site: bool mask} replaces the ReLU
    decision of that site by the given mask (see _relu_ties)."""
    from oracle import cheb_oracle as O
    B = x.shape[0]
    torch.set_num_threads(min(16, os.cpu_count() or 16))    # (the GPU box shares its host: 16 cores per GPU)

    class Recording(O.OracleVAE):
        def _relu(self, site, t):
            self.pre[site] = t.detach()
            if pins and site in pins:
                return t * pins[site].to(t.dtype)
            return torch.relu(t)
    ora = Recording(dict(cfg, dropout=P_DROP), O.Topology(np.load(os.path.join(ROOT, "tests", "golden", topo_name))),
                    {k: v.cpu() for k, v in net.state_dict().items()}, requires_grad=True, dtype=dtype)
    ora.pre = {}
    ora.training = True
    x, eps = x.to(dtype), eps.to(dtype)
    blocks = _drop_blocks(drop_u, B, H, flat)

    def masked_drop(t):          # F.dropout's arithmetic on the shared uniforms, in the model's call order
        u = blocks.pop(0)
        assert u.shape == t.shape
        return torch.where(u >= P_DROP, t / (1.0 - P_DROP), torch.zeros_like(t))
    ora._drop = masked_drop
    lo, co, ro, (ko, reco, zo), yo, _, _ = ora.forward(x, x.clone(), y.to(dtype), "train", eps=eps)
    lo.backward()
    assert not blocks
    return dict(loss=float(lo.detach()), correct=int(co), recon=ro.detach(), kld=ko.detach(), rec=reco.detach(),
                z=zo.detach(), y_hat=yo.detach(), grads=ora.grads(), pre=ora.pre)


def _relu_ties(nat, net, want, drop_u, tag, topo_name, max_ties=4):
    """Two fp32 evaluation orders disagree on the sign of a pre-activation that is zero to rounding (measured: B = 64
    meshes of the 5k model have 15 M ReLU inputs of magnitude O(1); about one per step lies within 1e-7 of zero, e.g.
    2.5e-8 in float64), and the ReLU derivative is discontinuous there, so ONE such element moves every upstream
    gradient by 1e-5 .. 1e-4 relative.  That is not an error of either side.  This helper compares the ReLU decisions
    of the native step (the stored post-activation tensors of its workspace) with the oracle's at every conv site and
    dense site, REQUIRES every disagreement to sit at |pre-activation| <= 5e-6 of the site's largest -- 20 x tighter than the
    forward bar; K = 10 recurrences at the 20k level carry 1e-6 of fp32 noise -- (anything else
    is a real sign error and fails), and returns pins = {site: this library's mask} for the sites that have ties."""
    n, B = net.n_layers, nat.B
    H, flat = net.num_hidden, net.dec_lin_2.out_features
    blocks = _drop_blocks(drop_u, B, H, flat)
    # encoder convs: only the rows the one-hot pooling selects exist (and matter: the ReLU of an un-selected row feeds
    # nothing) -- the step stores the pooled tensor encP[i] = relu(pre)[:, D_i.col]
    topo = np.load(os.path.join(ROOT, "tests", "golden", topo_name))
    sites = [(f"cheb.{i}", ("encP", i), None) for i in range(n)] + [(f"cheb_dec.{i}", ("decC", i), None) for i in range(n)]
    sites += [("enc_lin", ("h", 0), blocks[0]), ("dec_lin", ("d1", 0), blocks[2]), ("dec_lin_2", ("d2", 0), blocks[3])]
    pins, notes = {}, []
    for site, (name, idx), u in sites:
        pre = want["pre"][site]
        theirs = pre > 0
        if name == "encP":
            sel = torch.from_numpy(topo[f"D{idx}_col"].astype(np.int64))
            ours = theirs.clone()
            ours[:, sel] = nat.ws_tensor(name, idx).cpu().reshape(pre.shape[0], len(sel), pre.shape[2]) > 0
        else:
            ours = nat.ws_tensor(name, idx).cpu().reshape(pre.shape) > 0
        diff = ours != theirs
        if u is not None:                       # a dropped element says nothing about the ReLU decision
            diff &= (u >= P_DROP)
        if not bool(diff.any()):
            continue
        scale = float(pre.abs().max())
        worst = float(pre[diff].abs().max())
        assert worst <= 5e-6 * scale, (site, "ReLU sign differs at a pre-activation that is NOT a tie", worst, scale)
        pins[site] = torch.where(diff, ours, theirs)
        notes.append(f"{site}: {int(diff.sum())} tie(s), |pre| <= {worst:.1e} of max {scale:.1e}")
    n_ties = sum(int(x.split(": ")[1].split(" ")[0]) for x in notes)
    print(f"[{tag}] ReLU ties: {n_ties} (bound {max_ties})" + (" -- pinned to this library's decision: " + "; ".join(notes) if notes else ""))
    # the pinning bends the checker towards the product, so its use is counted: measured ~1 per 5k step at B = 64 (15 M ReLU
    # inputs), a few dozen per 20k step; a kernel that started to disagree with the reference beyond rounding would show here
    assert n_ties <= max_ties, (tag, n_ties, notes)
    return pins


def _compare_with_oracle(got, want, tag, grad_bar=1e-4, truth=None):
    torch.testing.assert_close(got["z"], want["z"], rtol=0, atol=FWD_ATOL)
    torch.testing.assert_close(got["y_hat"], want["y_hat"], rtol=0, atol=FWD_ATOL)
    torch.testing.assert_close(got["recon"], want["recon"], rtol=0, atol=FWD_ATOL)       # ALL meshes
    torch.testing.assert_close(got["kld"], want["kld"], rtol=1e-5, atol=FWD_ATOL)
    torch.testing.assert_close(got["rec"].float(), want["rec"].float(), rtol=2e-6, atol=1e-2)
    assert got["correct"] == want["correct"]
    assert abs(got["loss"] - want["loss"]) <= 2e-6 * abs(want["loss"]) + 1e-2
    per_mesh = (got["recon"] - want["recon"]).abs().flatten(1).max(dim=1).values
    worst, worst_k = 0.0, None
    assert set(want["grads"]) <= set(got["grads"])        # (dec_lin_1 has no gradient in the reference: its span stays zero)
    assert all(float(got["grads"][k].abs().max()) == 0.0 for k in set(got["grads"]) - set(want["grads"]))
    table, bad = [], []
    for k, gref in want["grads"].items():
        rel = float((got["grads"][k] - gref).norm()) / max(float(gref.norm()), 1e-12)
        row = f"{k}={rel:.1e}"
        if truth is not None:        # distance of this library / of the reference's fp32 arithmetic from the exact answer
            t = truth["grads"][k]
            row += f"({float((got['grads'][k].double() - t).norm() / t.norm()):.1e}/{float((gref.double() - t).norm() / t.norm()):.1e})"
        table.append(row)
        if rel > worst:
            worst, worst_k = rel, k
        if not rel < grad_bar:
            bad.append((k, rel))
    print(f"[{tag}] gradient rel error vs oracle" + (" (vs fp64 truth: this library / reference fp32)" if truth else "") + ": " + " ".join(table))
    assert not bad, bad
    print(f"[{tag}] max|recon - oracle| per mesh: worst {float(per_mesh.max()):.2e} (mesh {int(per_mesh.argmax())}); "
          f"worst relative gradient error {worst:.2e} ({worst_k})")


def test_b64_fp32_step_matches_oracle_on_all_meshes():
    """configs[1]'s workload as bench.py times it (5k template, K = 6, B = 64, dropout on, fwd + bwd), fp32."""
    dev = _dev()
    B = 64
    net = _build(CFG_5K, "topology_5k.npz", dev).train()
    x, y, eps, g = _inputs(net, B)
    H, flat = net.num_hidden, net.dec_lin_2.out_features
    drop_u = torch.rand(B * (3 * H + flat), generator=g)
    nat, got = _native(net, B, x, y, eps, drop_u)
    assert nat.u_cols == 3 * H + flat
    want = _oracle(CFG_5K, "topology_5k.npz", net, x, y, eps, drop_u, H, flat)
    pins = _relu_ties(nat, net, want, drop_u, "b64 fp32 5k", "topology_5k.npz")
    if pins:
        want = _oracle(CFG_5K, "topology_5k.npz", net, x, y, eps, drop_u, H, flat, pins=pins)
    truth = _oracle(CFG_5K, "topology_5k.npz", net, x, y, eps, drop_u, H, flat, dtype=torch.float64, pins=pins)
    _compare_with_oracle(got, want, "b64 fp32 5k", truth=truth)


def test_b64_bf16_step_against_fp32_step_and_oracle():
    """The bf16-storage step at B = 64 against the fp32 HIP step on the same inputs and masks (bars of
    test_gpu_bf16.py: recon within 1e-2 of max|recon|, z 1e-3, loss 1e-5 relative, gradients cosine > 0.99 and 0.2 /
    3e-2 relative), mesh by mesh -- the l0h kernels' block -> mesh map above B = 8."""
    dev = _dev()
    B = 64
    net = _build(CFG_5K, "topology_5k.npz", dev).train()
    x, y, eps, g = _inputs(net, B)
    H, flat = net.num_hidden, net.dec_lin_2.out_features
    drop_u = torch.rand(B * (3 * H + flat), generator=g)
    _, f32 = _native(net, B, x, y, eps, drop_u, storage="f32")
    _, b16 = _native(net, B, x, y, eps, drop_u, storage="bf16")
    assert not torch.equal(f32["recon"], b16["recon"])
    rscale = float(f32["recon"].abs().max())
    e_mesh = (b16["recon"] - f32["recon"]).abs().flatten(1).max(dim=1).values / rscale
    assert float(e_mesh.max()) < 1e-2, (int(e_mesh.argmax()), float(e_mesh.max()))       # every mesh, not a slice
    assert float((b16["z"] - f32["z"]).abs().max()) < 1e-3
    assert abs(b16["loss"] - f32["loss"]) < 1e-5 * abs(f32["loss"])
    worst, worst_k = 0.0, None
    for k, gref in f32["grads"].items():
        if float(gref.abs().max()) == 0.0:          # dec_lin_1: no gradient in the reference, the span stays zero
            assert float(b16["grads"][k].abs().max()) == 0.0
            continue
        rel = float((b16["grads"][k] - gref).norm()) / max(float(gref.norm()), 1e-12)
        cos = float(torch.nn.functional.cosine_similarity(b16["grads"][k].reshape(1, -1).double(), gref.reshape(1, -1).double()))
        assert cos > 0.99, (k, cos)
        # (test_gpu_bf16.GRAD_BARS: 2 x the per-tensor figures measured at B = 4; a batch of 64 averages the storage noise
        #  of 16 x the meshes -- measured worst 2.0e-2 (cheb.0.weight) against 6.0e-2 at B = 4 -- so half of those bars)
        assert rel < grad_bar("5k", k, 0.5), (k, rel)
        if rel > worst:
            worst, worst_k = rel, k
    print(f"[b64 bf16 5k] recon vs fp32 step: worst mesh {float(e_mesh.max()):.2e} of max|recon|; worst gradient rel {worst:.2e} ({worst_k})")


def test_b64_hires20k_step_matches_oracle_on_all_meshes():
    """BASELINE configs[3] at B = 64: meshes 8..63 of cheb_big's (mesh, channel pair) map, k_reduce_partials with
    G >= 128 (rows >= 131 072), k_spmm's mesh -> XCD tiling -- all against the CPU oracle, all meshes."""
    dev = _dev()
    B = 64
    net = _build(CFG_20K, "topology_20k.npz", dev).train()
    x, y, eps, g = _inputs(net, B, seed=32)
    H, flat = net.num_hidden, net.dec_lin_2.out_features
    drop_u = torch.rand(B * (3 * H + flat), generator=g)
    nat, got = _native(net, B, x, y, eps, drop_u)
    want = _oracle(CFG_20K, "topology_20k.npz", net, x, y, eps, drop_u, H, flat)
    pins = _relu_ties(nat, net, want, drop_u, "b64 fp32 20k", "topology_20k.npz", max_ties=120)
    if pins:
        want = _oracle(CFG_20K, "topology_20k.npz", net, x, y, eps, drop_u, H, flat, pins=pins)
    # (the float64 run beside it -- this library ~1e-5, the reference's fp32 1-2e-6 from the exact answer at this size --
    #  is printed by the 5k test; here it would add a third 20k oracle pass to a test that already takes two minutes)
    _compare_with_oracle(got, want, "b64 fp32 20k")


@pytest.mark.parametrize("which", ["5k", "20k"])
def test_b64_equals_eight_b8_runs_bitwise(which):
    """B = 64 == the concatenation of eight B = 8 runs, bit for bit: level-0 conv forward and dX through the module API
    (3 -> 16, 16 -> 16, the 16 -> 3 final-layer quirk), and every per-mesh output of the whole native step (recon, z,
    y_hat, kld, rec) with the noise and the dropout uniforms of the chunk."""
    dev = _dev()
    cfg, topo = (CFG_5K, "topology_5k.npz") if which == "5k" else (CFG_20K, "topology_20k.npz")
    net = _build(cfg, topo, dev).train()
    net._prepare()
    B, b = 64, 8
    g = torch.Generator().manual_seed(5)
    N = net.num_nodes[0]
    n = net.n_layers
    layers = [("cheb.0", net.cheb[0], 0, net.filters[0]), (f"cheb_dec.{n - 1}", net.cheb_dec[n - 1], 0, net.filters[1]),
              (f"cheb_dec.{n}", net.cheb_dec[n], -1, net.filters[1])]
    for name, conv, lvl, cin in layers:
        ei, nrm = net.A_edge_index[lvl], net.A_norm[lvl]
        xin = torch.randn(B, N, cin, generator=g).to(dev)
        gout = torch.randn(B, N, conv.out_channels, generator=g).to(dev)
        xa = xin.clone().requires_grad_(True)
        oa = conv(xa, ei, nrm)
        oa.backward(gout)
        for c in range(B // b):
            xc = xin[c * b:(c + 1) * b].clone().requires_grad_(True)
            oc = conv(xc, ei, nrm)
            oc.backward(gout[c * b:(c + 1) * b].clone())
            assert torch.equal(oc, oa[c * b:(c + 1) * b]), (name, "forward", c)
            assert torch.equal(xc.grad, xa.grad[c * b:(c + 1) * b]), (name, "dX", c)
    x, y, eps, g = _inputs(net, B, seed=33)
    H, flat = net.num_hidden, net.dec_lin_2.out_features
    drop_u = torch.rand(B * (3 * H + flat), generator=g)
    _, full = _native(net, B, x, y, eps, drop_u)
    blocks = _drop_blocks(drop_u, B, H, flat)
    for c in range(B // b):
        sl = slice(c * b, (c + 1) * b)
        du = torch.cat([blk[sl].reshape(-1) for blk in blocks])
        _, part = _native(net, b, x[sl].contiguous(), y[sl].contiguous(), eps[sl].contiguous(), du)
        for k in ("recon", "z", "y_hat", "kld", "rec"):
            assert torch.equal(part[k], full[k][sl]), (k, c)


@pytest.mark.parametrize("B", [1, 7, 33, 63, 65, 96, 128])
def test_ragged_batch_sizes_match_oracle(B):
    """Batches that are not multiples of the 8 XCDs (the block -> mesh maps end in `if (mesh >= B) return`), a single
    mesh, one mesh more than the benchmarked batch, and batches of one and a half and two rounds of the 5k level's
    one-workgroup-per-CU kernels (96, 128 meshes = 384, 512 patch workgroups): the whole fp32 step against the oracle,
    all meshes, same bars."""
    dev = _dev()
    net = _build(CFG_5K, "topology_5k.npz", dev).train()
    x, y, eps, g = _inputs(net, B, seed=40 + B)
    H, flat = net.num_hidden, net.dec_lin_2.out_features
    drop_u = torch.rand(B * (3 * H + flat), generator=g)
    nat, got = _native(net, B, x, y, eps, drop_u)
    want = _oracle(CFG_5K, "topology_5k.npz", net, x, y, eps, drop_u, H, flat)
    pins = _relu_ties(nat, net, want, drop_u, f"b{B} fp32 5k", "topology_5k.npz")
    if pins:
        want = _oracle(CFG_5K, "topology_5k.npz", net, x, y, eps, drop_u, H, flat, pins=pins)
    _compare_with_oracle(got, want, f"b{B} fp32 5k")
