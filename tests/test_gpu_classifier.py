"""GPU (-m gpu): the crecon classifier path -- cheb_GCN on libmeshvae_hip's
ChebConv / pool / linear kernels and crecon's estimate_diff -- against vectors captured from the
reference (oracle/make_golden_cls.py) and against the CPU oracle on fresh inputs.

Tolerances: forward 1e-4 absolute; gradients 1e-4 relative per tensor (fixed summation order differs).
"""
import os

import numpy as np
import pytest
import torch

from conftest import CFG_5K, ROOT, TINY_CFG, state_dict_from

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _t(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def _classifier(which, dev):
    from model import load_topology
    from models.cheb_cls import cheb_GCN
    cfg, topo = (TINY_CFG, "topology_tiny.npz") if which == "tiny" else (CFG_5K, "topology_5k.npz")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", topo), dev)
    torch.manual_seed(666)
    return cheb_GCN(6, dict(cfg, num_conv_filters=list(cfg["num_conv_filters"])), D, U, A, nn_).to(dev)


@pytest.mark.parametrize("which", ["tiny", "5k"])
def test_classifier_train_step_matches_reference(which, cls_tiny_npz, cls_5k_npz):
    npz = cls_tiny_npz if which == "tiny" else cls_5k_npz
    dev = _dev()
    net = _classifier(which, dev)
    assert all(torch.equal(v.cpu(), state_dict_from(npz)[k]) for k, v in net.state_dict().items())
    net.train()
    logits = net(_t(npz["x"], dev))
    loss = torch.nn.CrossEntropyLoss()(logits, _t(npz["label"], dev))       # crecon.py:83,262
    loss.backward()
    torch.testing.assert_close(logits.detach().cpu(), _t(npz["logits"]), rtol=0, atol=1e-4)
    torch.testing.assert_close(loss.detach().cpu(), _t(npz["loss"]), rtol=1e-5, atol=1e-5)
    got = {k: p.grad for k, p in net.named_parameters()}
    assert sorted(got) == sorted(str(k) for k in npz["grad_names"])
    worst = 0.0
    for k, g in got.items():
        want = _t(npz[f"grad/{k}"])
        assert g is not None and g.shape == want.shape, k
        rel = (g.cpu() - want).norm().item() / max(want.norm().item(), 1e-12)
        worst = max(worst, rel)
        assert rel < 1e-4, (k, rel)
    print(f"[{which}] classifier worst relative gradient error = {worst:.3e}")


def test_classifier_matches_oracle_on_fresh_input(topo5k_npz, cls_5k_npz):
    """B = 7 (odd), eval mode, weights perturbed away from the seed: HIP vs the CPU restatement."""
    from oracle import cheb_oracle as O
    dev = _dev()
    net = _classifier("5k", dev)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in net.parameters():
            p.add_((torch.randn(p.shape, generator=g) * 0.05).to(dev))
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    x = torch.randn(7, 4998, 6, generator=g)
    want = O.OracleGCN(CFG_5K, O.Topology(topo5k_npz), sd).forward(x)
    net.eval()
    with torch.no_grad():
        got = net(x.to(dev).reshape(7, -1))                  # flat input: forward reshapes (cheb_cls.py:92)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-4)


def test_pyg_chebconv_module_shapes_and_padding(topotiny_npz, ops_npz):
    """ChebConv alone: [N, C] input, K = 1, and the zero-padded 5/6/7-channel inputs against the oracle."""
    from models.cheb_cls import ChebConv
    from oracle import cheb_oracle as O
    dev = _dev()
    ei = _t(np.vstack([topotiny_npz["A0_row"], topotiny_npz["A0_col"]]).astype(np.int64))
    N = int(topotiny_npz["num_nodes"][0])
    g = torch.Generator().manual_seed(11)
    for cin, cout, K in ((6, 16, 6), (5, 8, 3), (7, 12, 2), (6, 4, 1), (3, 16, 6), (16, 32, 4)):
        torch.manual_seed(cin * 100 + cout)
        conv = ChebConv(cin, cout, K).to(dev)
        with torch.no_grad():
            conv.bias.copy_(torch.randn(cout, generator=g))
        x = torch.randn(3, N, cin, generator=g)
        lins = [l.weight.detach().cpu() for l in conv.lins]
        want = O.pyg_cheb_conv(x, ei, lins, conv.bias.detach().cpu())
        xg = x.to(dev).requires_grad_(True)
        got = conv(xg, ei.to(dev))
        torch.testing.assert_close(got.detach().cpu(), want, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(conv(xg[0], ei.to(dev)).detach().cpu(), want[0], rtol=1e-4, atol=1e-4)
        # gradients through the stacked / padded weights reach every per-order Linear and the input
        xo = x.clone().requires_grad_(True)
        lo = [w.clone().requires_grad_(True) for w in lins]
        gy = torch.randn(3, N, cout, generator=g)
        O.pyg_cheb_conv(xo, ei, lo, conv.bias.detach().cpu()).backward(gy)
        got.backward(gy.to(dev))
        torch.testing.assert_close(xg.grad.cpu(), xo.grad, rtol=1e-3, atol=1e-4)
        for k in range(K):
            torch.testing.assert_close(conv.lins[k].weight.grad.cpu(), lo[k].grad, rtol=1e-3, atol=2e-4)
        torch.testing.assert_close(conv.bias.grad.cpu(), gy.sum((0, 1)), rtol=1e-4, atol=1e-4)
    with pytest.raises(NotImplementedError):
        ChebConv(3, 4, 2).to(dev)(torch.zeros(1, N, 3, device=dev), ei.to(dev), lambda_max=3.0)


def test_estimate_diff_matches_reference(cls_5k_npz, model_5k_npz):
    """crecon.py:160-198 on the seed-666 VAE (eval mode): both label modes, and the single-mesh call shape."""
    from crecon_ops import classifier_, estimate_diff
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = _dev()
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
    torch.manual_seed(666)
    vae = cheb_VAE(3, CFG_5K, D, U, A, nn_, model="optimal_sigma_VAE").to(dev)
    vae.load_state_dict(state_dict_from(model_5k_npz))
    vae.eval()
    x, label = _t(cls_5k_npz["diff/x"], dev), _t(cls_5k_npz["diff/label"], dev)
    for mode in ("train", "test"):
        diff, correct = estimate_diff(vae, x, label, mode)
        assert correct == int(cls_5k_npz[f"diff/{mode}_correct"]) and diff.shape == (4, 4998, 6)
        torch.testing.assert_close(diff.cpu(), _t(cls_5k_npz[f"diff/{mode}"]), rtol=0, atol=1e-4)
    one, c1 = estimate_diff(vae, x[2], int(label[2]), "train")
    torch.testing.assert_close(one.cpu()[0], _t(cls_5k_npz["diff/train"])[2], rtol=0, atol=1e-4)
    with torch.no_grad():
        pred = classifier_(vae, x)
    assert int((pred == label).sum()) == int(cls_5k_npz["diff/test_correct"])
    # end to end: diff -> classifier logits stay finite and differ between the two label modes
    net = _classifier("5k", dev).eval()
    with torch.no_grad():
        la = net(estimate_diff(vae, x, label, "train")[0])
        lb = net(estimate_diff(vae, x, label, "test")[0])
    assert torch.isfinite(la).all() and torch.isfinite(lb).all() and not torch.equal(la, lb)


def test_inference_hipgraph_replay_config4(cls_5k_npz, model_5k_npz):
    """BASELINE configs[4]: the VAE inference of crecon.py:170-192 (encoder -> classifier -> z_mean -> decoder for the
    predicted and the opposite label) captured in a hipGraph on the 5k template -- bench.py --config infer's own
    functions.  Replay == eager bitwise at B = 1 / 32 / 256 (and after the input buffer is refilled), and at the
    fixture's batch the replayed reconstructions reproduce the reference's `estimate_diff(..., "test")` vectors."""
    import bench
    dev = _dev()
    net = bench.build_model(dev)                       # seed-666 weights = the fixture's state_dict
    net.load_state_dict(state_dict_from(model_5k_npz))
    net.eval()
    fn = bench.estimate_diff_fn(net)
    with torch.no_grad():
        for B in (1, 32, 256):
            g0 = torch.Generator().manual_seed(B)
            x = torch.randn(B, 4998, 3, generator=g0).to(dev)
            ref = [t.clone() for t in fn(x)]
            graph, out = bench.capture_inference(fn, x, dev)
            graph.replay()
            torch.cuda.synchronize()
            for a, b in zip(out, ref):
                assert torch.equal(a, b), B
            x.copy_(torch.randn(B, 4998, 3, generator=g0))            # new meshes through the same graph
            graph.replay()
            torch.cuda.synchronize()
            for a, b in zip(out, fn(x)):
                assert torch.equal(a, b), B
        # the reference's vectors (oracle/make_golden_cls.py: crecon.estimate_diff on the seed-666 VAE, eval mode, B = 4)
        x4 = torch.from_numpy(cls_5k_npz["diff/x"]).to(dev)
        graph, (recon, recon_oppo, y_hat) = bench.capture_inference(fn, x4, dev)
        graph.replay()
        torch.cuda.synchronize()
        diff = torch.cat((x4 - recon_oppo, x4 - recon), dim=-1).cpu()
        want = torch.from_numpy(cls_5k_npz["diff/test"])
        assert float((diff - want).abs().max()) < 1e-4, float((diff - want).abs().max())
        label = torch.from_numpy(cls_5k_npz["diff/label"])
        assert int((y_hat.argmax(-1).cpu() == label).sum()) == int(cls_5k_npz["diff/test_correct"])


def test_classifier_step_graph_equals_eager_and_module_path():
    """engine.ClassifierStep: the hipGraph replay is bitwise equal to the eager launch sequence, and both follow
    the plain module path (estimate_diff -> cheb_GCN -> CrossEntropyLoss -> torch.optim.Adam, crecon.py:65-100)."""
    from crecon_ops import estimate_diff
    from meshvae_hip.engine import ClassifierStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = _dev()
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), dev)
    torch.manual_seed(666)
    vae = cheb_VAE(3, TINY_CFG, D, U, A, nn_, model="optimal_sigma_VAE").to(dev).eval()   # deterministic VAE
    B = 8
    g = torch.Generator().manual_seed(2)
    xs = [torch.randn(B, nn_[0], 3, generator=g).to(dev) for _ in range(4)]
    ys = [torch.randint(0, 2, (B,), generator=g).to(dev) for _ in range(4)]

    def run(mode):
        net = _classifier("tiny", dev)
        losses = []
        if mode == "module":
            opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
            for x, y in zip(xs, ys):
                diff, _ = estimate_diff(vae, x, y, "train")
                opt.zero_grad()
                loss = torch.nn.CrossEntropyLoss()(net(diff), y)
                loss.backward()
                opt.step()
                losses.append(loss.detach().clone())
        else:
            cs = ClassifierStep(net, vae, B, lr=1e-3, weight_decay=5e-4, use_graph=(mode == "graph"))
            if mode == "graph":
                cs.load(xs[0], ys[0])
                cs.capture()                       # warm-up steps do not touch the weights (no optimizer inside)
            for x, y in zip(xs, ys):
                cs.load(x, y)
                loss, logits, vae_correct = cs.step()
                assert logits.shape == (B, 2) and 0 <= int(vae_correct) <= B
                losses.append(loss.clone())
        return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}

    l_e, sd_e = run("eager")
    l_g, sd_g = run("graph")
    l_m, sd_m = run("module")
    assert torch.equal(l_e, l_g) and all(torch.equal(sd_e[k], sd_g[k]) for k in sd_e)
    torch.testing.assert_close(l_e, l_m, rtol=1e-5, atol=1e-6)
    for k in sd_m:
        torch.testing.assert_close(sd_e[k], sd_m[k], rtol=1e-4, atol=2e-5, msg=k)
    assert not torch.equal(l_e[0], l_e[-1])
