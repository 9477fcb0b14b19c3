"""CPU: pin the oracle (oracle/cheb_oracle.py) against vectors captured from the reference
itself (oracle/make_golden.py).  Tolerances: the oracle restates the same op sequence, so
agreement is at fp32 rounding level (1e-6 relative)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import CFG_5K, CFG_20K, TINY_CFG, load_golden, state_dict_from
from oracle import cheb_oracle as O


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_topology_hashes_match_survey_probe(topo5k_npz):
    # SURVEY.md section 8(c) probe hashes of the reference-generated hierarchy
    want = ["392d175745f568af", "e69d8c2b28c39335", "b9e6a3ede1512b5f", "2e8b03dcdc4feec8", "ea634efafdf9a9c6"]
    for i, w in enumerate(want):
        a = np.vstack([topo5k_npz[f"A{i}_row"], topo5k_npz[f"A{i}_col"]]).astype(np.int64)
        assert _sha16(a) == w
    assert list(topo5k_npz["num_nodes"]) == [4998, 1250, 313, 79, 20]
    assert _sha16(topo5k_npz["D0_col"].astype(np.int64)) == "613c3bdbdee65086"
    for i in range(4):
        assert np.all(topo5k_npz[f"D{i}_val"] == 1.0)
        assert len(topo5k_npz[f"U{i}_row"]) == 3 * topo5k_npz["num_nodes"][i]


@pytest.mark.parametrize("which", ["5k", "tiny"])
def test_cheb_norm(which, topo5k_npz, topotiny_npz):
    npz = topo5k_npz if which == "5k" else topotiny_npz
    for i in range(int(npz["n_levels"])):
        ei = _t(np.vstack([npz[f"A{i}_row"], npz[f"A{i}_col"]]).astype(np.int64))
        ei2, nrm = O.cheb_norm(ei, int(npz["num_nodes"][i]))
        assert torch.equal(ei2, ei)
        assert torch.equal(nrm, _t(npz[f"A{i}_norm"]))
    if which == "5k":
        assert abs(float(topo5k_npz["A0_norm"][0]) + 1 / np.sqrt(42)) < 1e-7


def test_cheb_conv_cases(ops_npz, topotiny_npz):
    for case in ops_npz["case_names"]:
        level, n_x, B, cin, cout, K, has_b = [int(v) for v in ops_npz[f"{case}_meta"]]
        ei = _t(np.vstack([topotiny_npz[f"A{level}_row"], topotiny_npz[f"A{level}_col"]]).astype(np.int64))
        ei, nrm = O.cheb_norm(ei, int(topotiny_npz["num_nodes"][level]))
        x = _t(ops_npz[f"{case}_x"]).requires_grad_(True)
        w = _t(ops_npz[f"{case}_w"]).requires_grad_(True)
        b = _t(ops_npz[f"{case}_b"]).requires_grad_(True) if has_b else None
        y = O.cheb_conv(x, ei, nrm, w, b)
        torch.testing.assert_close(y, _t(ops_npz[f"{case}_y"]), rtol=1e-6, atol=1e-6)
        y.backward(_t(ops_npz[f"{case}_gy"]))
        torch.testing.assert_close(x.grad, _t(ops_npz[f"{case}_gx"]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(w.grad, _t(ops_npz[f"{case}_gw"]), rtol=1e-5, atol=1e-5)
        if has_b:
            torch.testing.assert_close(b.grad, _t(ops_npz[f"{case}_gb"]), rtol=1e-5, atol=1e-5)


def test_quirk_closed_form(ops_npz):
    """cheb_VAE.py:288: rows >= 11 (coarsest level size) see Tx1=0, Tx2=-x, Tx3=0, Tx4=x, Tx5=0."""
    x, w, y = (_t(ops_npz[f"c_quirk_16_3_k6_{k}"]) for k in ("x", "w", "y"))
    closed = x[:, 11:] @ (w[0] - w[2] + w[4])
    torch.testing.assert_close(y[:, 11:], closed, rtol=1e-5, atol=1e-5)


def test_surface_pool_cases(ops_npz, topotiny_npz):
    topo = O.Topology(topotiny_npz)
    for tag, mat in (("p_D0", topo.D[0]), ("p_D1", topo.D[1]), ("p_U0", topo.U[0]), ("p_U1", topo.U[1])):
        x = _t(ops_npz[f"{tag}_x"]).requires_grad_(True)
        y = O.surface_pool(x, *mat)
        assert torch.equal(y, _t(ops_npz[f"{tag}_y"]))          # same op sequence -> bit-exact
        y.backward(_t(ops_npz[f"{tag}_gy"]))
        torch.testing.assert_close(x.grad, _t(ops_npz[f"{tag}_gx"]), rtol=1e-6, atol=1e-6)
    # D is a pure row gather
    x = _t(ops_npz["p_D0_x"])
    assert torch.equal(_t(ops_npz["p_D0_y"]), x[:, topo.D[0][0][1]])
    with pytest.raises(ValueError):
        O.surface_pool(x[:, :-1], *topo.D[0])


def test_logpdf(ops_npz):
    assert torch.equal(O.kld(_t(ops_npz["kld_mu"]), _t(ops_npz["kld_lv"])), _t(ops_npz["kld_out"]))
    ls = O.softclip(torch.Tensor([1]), -6)
    assert torch.equal(ls, _t(ops_npz["log_sigma"]))
    assert abs(O.log_sigma_const() - float(ls)) < 1e-6
    assert torch.equal(O.gaussian_nll(_t(ops_npz["nll_mu"]), ls, _t(ops_npz["nll_x"])), _t(ops_npz["nll_out"]))


@pytest.mark.parametrize("which", ["tiny", "5k"])
def test_init_order_matches_reference(which, model_tiny_npz, model_5k_npz, topotiny_npz, topo5k_npz):
    npz, tnpz, cfg = (model_tiny_npz, topotiny_npz, TINY_CFG) if which == "tiny" else (model_5k_npz, topo5k_npz, CFG_5K)
    torch.manual_seed(666)
    sd = O.init_state_dict(cfg, O.Topology(tnpz))
    want = state_dict_from(npz)
    assert list(sd.keys()) != [] and set(sd.keys()) == set(want.keys())
    for k in want:
        assert torch.equal(sd[k], want[k]), k
    if which == "5k":
        assert len(want) == 31 and sum(v.numel() for v in want.values()) == 712642


@pytest.mark.parametrize("which", ["tiny", "5k"])
def test_full_model_eval_and_train(which, model_tiny_npz, model_5k_npz, topotiny_npz, topo5k_npz):
    npz, tnpz, cfg = (model_tiny_npz, topotiny_npz, TINY_CFG) if which == "tiny" else (model_5k_npz, topo5k_npz, CFG_5K)
    topo = O.Topology(tnpz)
    sd = state_dict_from(npz)
    x, y = _t(npz["x"]), _t(npz["y"])
    net = O.OracleVAE(cfg, topo, sd)
    with torch.no_grad():
        loss, correct, recon, (k, rec, z_), y_hat, mu, logvar = net.forward(x, x.clone(), y, "test")
        loss64, _, _, (_, rec64, _), _, _, _ = net.forward(x, x.double(), y, "test")
        oppo = net.sample(1 - y, z_)
    tol = dict(rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(y_hat, _t(npz["eval/y_hat"]), **tol)
    torch.testing.assert_close(mu, _t(npz["eval/mu"]), **tol)
    torch.testing.assert_close(logvar, _t(npz["eval/logvar"]), **tol)
    torch.testing.assert_close(k, _t(npz["eval/kld"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(recon, _t(npz["eval/recon"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(oppo, _t(npz["eval/oppo_recon"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rec, _t(npz["eval/rec"]), rtol=1e-6, atol=1e-2)
    torch.testing.assert_close(loss, _t(npz["eval/loss"]), rtol=1e-6, atol=1e-2)
    assert rec64.dtype == torch.float64 and loss64.dtype == torch.float64   # SURVEY section 7 "fp64 loss"
    torch.testing.assert_close(rec64, _t(npz["eval/rec64"]), rtol=1e-9, atol=1e-4)
    assert int(correct) == int(npz["eval/correct"])

    cfg0 = dict(cfg, dropout=0.0)
    net = O.OracleVAE(cfg0, topo, sd, requires_grad=True)
    net.training = True
    torch.manual_seed(123)
    loss, correct, recon, (k, rec, z_), y_hat, mu, logvar = net.forward(x, x.clone(), y, "train")
    loss.backward()
    torch.testing.assert_close(z_, _t(npz["train/z"]), **tol)
    torch.testing.assert_close(loss, _t(npz["train/loss"]), rtol=1e-6, atol=1e-2)
    g = net.grads()
    assert set(g.keys()) == set(str(n) for n in npz["train/grad_names"])
    assert "dec_lin_1.weight" not in g                         # never used (cheb_VAE.py:165)
    for name, grad in g.items():
        torch.testing.assert_close(grad, _t(npz[f"train/grad/{name}"]), rtol=1e-4, atol=1e-5, msg=name)


def test_hires_20k_config_oracle_matches_reference(topo20k_npz, model_20k_npz):
    """BASELINE configs[3] (19 992-vertex template, 6 levels, K = 10): constructor RNG order, eval
    forward and the train-mode loss/gradients of the oracle against vectors captured from the
    reference (oracle/make_golden_20k.py); weights are not stored, they come from the seed."""
    npz = model_20k_npz
    assert list(topo20k_npz["num_nodes"]) == [19992, 4998, 1250, 313, 79, 20]
    topo = O.Topology(topo20k_npz)
    torch.manual_seed(666)
    sd = O.init_state_dict(CFG_20K, topo)
    assert [str(k) for k in npz["sd_keys"]] == list(sd.keys())
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - float(npz["sd_abs_sum"])) < 1e-6 * float(npz["sd_abs_sum"])
    assert torch.equal(sd["cheb.0.weight"].reshape(-1)[:64], _t(npz["sd_head/cheb.0.weight"]))
    assert torch.equal(sd["dec_lin_2.weight"].reshape(-1)[:64], _t(npz["sd_head/dec_lin_2.weight"]))
    B = 2
    x = torch.randn(B, 19992, 3, generator=torch.Generator().manual_seed(0))
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, num_classes=2)
    net = O.OracleVAE(CFG_20K, topo, sd)
    with torch.no_grad():
        loss, correct, recon, (k, rec, z_), y_hat, mu, logvar = net.forward(x, x.clone(), y, "test")
    torch.testing.assert_close(y_hat, _t(npz["eval/y_hat"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(z_, _t(npz["eval/z"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(k, _t(npz["eval/kld"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(recon[:, :512], _t(npz["eval/recon_head"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(recon[:, -512:], _t(npz["eval/recon_tail"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss, _t(npz["eval/loss"]), rtol=1e-6, atol=5e-2)
    net = O.OracleVAE(dict(CFG_20K, dropout=0.0), topo, sd, requires_grad=True)
    net.training = True
    torch.manual_seed(123)
    loss, *_ = net.forward(x, x.double(), y, "train")
    loss.backward()
    assert loss.dtype == torch.float64
    torch.testing.assert_close(loss.detach(), _t(npz["train/loss"]), rtol=1e-9, atol=1e-3)
    g = net.grads()
    assert set(g) == set(str(n) for n in npz["train/grad_names"])
    for name, grad in g.items():
        gn = float(npz[f"train/gnorm/{name}"])
        assert abs(float(grad.double().norm()) - gn) <= 1e-4 * gn + 1e-7, name
        torch.testing.assert_close(grad.reshape(-1)[:1024], _t(npz[f"train/grad_head/{name}"]), rtol=1e-3,
                                   atol=1e-4 * max(gn / max(grad.numel() ** 0.5, 1.0), 1e-3), msg=name)


# --------------------------------------------------------------------------- crecon classifier
@pytest.mark.parametrize("which", ["tiny", "5k"])
def test_classifier_oracle_matches_reference(which, cls_tiny_npz, cls_5k_npz, topotiny_npz, topo5k_npz):
    """cheb_GCN (models/cheb_cls.py) run by the reference over refshim's PyG-2.0.4 ChebConv: seed-666
    initial weights bit-exact (names, order, values), logits / CE loss / all gradients at fp32 rounding."""
    npz, cfg, topo = ((cls_tiny_npz, TINY_CFG, O.Topology(topotiny_npz)) if which == "tiny"
                      else (cls_5k_npz, CFG_5K, O.Topology(topo5k_npz)))
    torch.manual_seed(666)
    sd = O.gcn_init_state_dict(cfg, topo)
    want = state_dict_from(npz)
    assert list(sd.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(sd[k], want[k]), k
    net = O.OracleGCN(cfg, topo, want, requires_grad=True)
    logits = net.forward(_t(npz["x"]))
    torch.testing.assert_close(logits, _t(npz["logits"]), rtol=1e-5, atol=1e-5)
    loss = torch.nn.functional.cross_entropy(logits, _t(npz["label"]))
    torch.testing.assert_close(loss, _t(npz["loss"]), rtol=1e-5, atol=1e-6)
    loss.backward()
    g = net.grads()
    assert sorted(g) == sorted(str(k) for k in npz["grad_names"])
    for k in g:
        w = _t(npz[f"grad/{k}"])
        torch.testing.assert_close(g[k], w, rtol=1e-4, atol=1e-5 * max(1.0, float(w.abs().max())))


def test_pyg_cheb_conv_standin_is_pinned_by_the_in_tree_operator(topotiny_npz):
    """cheb_pin.npz = the reference's OWN ChebConv (nn/conv.py:390-521, run unmodified by
    oracle/make_golden_cls_pin.py) on five shapes incl. K = 1, 2, no bias: refshim.PygChebConv -- the stand-in for
    torch-geometric's class that generated cls_*.npz -- and the oracle's pyg_cheb_conv must both reproduce its
    outputs and its autograd gradients, with weight[k] = lins.k.weight^T."""
    from oracle import refshim
    pin = load_golden("cheb_pin.npz")
    for tag in (str(c) for c in pin["cases"]):
        lvl, B, Cin, Cout, K, has_b = (int(v) for v in pin[f"{tag}/meta"])
        ei = torch.from_numpy(np.vstack([topotiny_npz[f"A{lvl}_row"], topotiny_npz[f"A{lvl}_col"]]).astype(np.int64))
        W = _t(pin[f"{tag}/weight"])                                   # in-tree layout [K, Cin, Cout]
        bias = _t(pin[f"{tag}/bias"]) if has_b else None
        shim = refshim.PygChebConv(Cin, Cout, K, bias=bool(has_b))
        with torch.no_grad():
            for k in range(K):
                shim.lins[k].weight.copy_(W[k].t())
            if has_b:
                shim.bias.copy_(bias)
        x = _t(pin[f"{tag}/x"]).requires_grad_(True)
        y = shim(x, ei)
        assert torch.equal(y.detach(), _t(pin[f"{tag}/y"])), tag        # same arithmetic, same order: bit-equal
        y.backward(_t(pin[f"{tag}/gy"]))
        torch.testing.assert_close(x.grad, _t(pin[f"{tag}/gx"]), rtol=1e-5, atol=1e-6, msg=tag)
        gw = torch.stack([lin.weight.grad.t() for lin in shim.lins])
        torch.testing.assert_close(gw, _t(pin[f"{tag}/gweight"]), rtol=1e-5, atol=1e-5, msg=tag)
        if has_b:
            torch.testing.assert_close(shim.bias.grad, _t(pin[f"{tag}/gbias"]), rtol=1e-5, atol=1e-5, msg=tag)
        # the oracle's functional restatement (what OracleGCN runs)
        lw = [W[k].t().contiguous().requires_grad_(True) for k in range(K)]
        x2 = _t(pin[f"{tag}/x"]).requires_grad_(True)
        y2 = O.pyg_cheb_conv(x2, ei, lw, bias)
        torch.testing.assert_close(y2, _t(pin[f"{tag}/y"]), rtol=1e-5, atol=1e-6, msg=tag)
        y2.backward(_t(pin[f"{tag}/gy"]))
        torch.testing.assert_close(x2.grad, _t(pin[f"{tag}/gx"]), rtol=1e-5, atol=1e-6, msg=tag)
        torch.testing.assert_close(torch.stack([w.grad.t() for w in lw]), _t(pin[f"{tag}/gweight"]), rtol=1e-5,
                                   atol=1e-5, msg=tag)


def test_estimate_diff_oracle_matches_reference(cls_5k_npz, model_5k_npz, topo5k_npz):
    """crecon.py:160-198 run by the reference on its seed-666 cheb_VAE (eval mode) vs the restatement."""
    vae = O.OracleVAE(CFG_5K, O.Topology(topo5k_npz), state_dict_from(model_5k_npz))
    x, label = _t(cls_5k_npz["diff/x"]), _t(cls_5k_npz["diff/label"])
    assert np.array_equal(cls_5k_npz["diff/x"], model_5k_npz["x"])
    for mode in ("train", "test"):
        diff, correct = O.estimate_diff(vae, x, label, mode)
        assert correct == int(cls_5k_npz[f"diff/{mode}_correct"])
        assert diff.shape == (4, 4998, 6)
        torch.testing.assert_close(diff, _t(cls_5k_npz[f"diff/{mode}"]), rtol=1e-5, atol=1e-5)
    assert not np.array_equal(cls_5k_npz["diff/train"], cls_5k_npz["diff/test"])   # labels differ from predictions


# --------------------------------------------------------------------------- SURVEY 8(f) next #2, input side
@pytest.mark.parametrize("tag", ["tiny", "5k"])
def test_procrustes_oracle_matches_reference(tag):
    """utils.procrustes run by the reference (oracle/make_golden_pre.py) vs the numpy restatement: same
    LAPACK / BLAS calls, so agreement is at the last bits (one case is a reflection, det R = -1)."""
    from conftest import load_golden
    npz = load_golden("procrustes.npz")
    for b, p in enumerate(npz[f"{tag}/pts"]):
        mtx1, mtx2, disparity, (R, s, m) = O.procrustes(npz[f"{tag}/template"], p)
        np.testing.assert_allclose(mtx1, npz[f"{tag}/mtx1"][b], rtol=0, atol=1e-15)
        np.testing.assert_allclose(mtx2, npz[f"{tag}/mtx2"][b], rtol=0, atol=1e-14)
        np.testing.assert_allclose(R, npz[f"{tag}/R"][b], rtol=0, atol=1e-13)
        np.testing.assert_allclose([disparity, s], [npz[f"{tag}/disparity"][b], npz[f"{tag}/s"][b]], rtol=1e-12)
        np.testing.assert_allclose(m, npz[f"{tag}/m"][b], rtol=1e-14)
    if tag == "tiny":
        assert np.linalg.det(npz["tiny/R"][2]) < 0
        with pytest.raises(ValueError, match="same shape"):
            O.procrustes(npz["tiny/template"], npz["tiny/pts"][0][:-1])
        with pytest.raises(ValueError, match="unique points"):
            O.procrustes(npz["tiny/template"], np.ones_like(npz["tiny/template"]))
