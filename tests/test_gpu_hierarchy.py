"""GPU box: the hierarchy generator (C++ libmeshvae_host.so under mesh-vae_amd/mesh_operations.py, SURVEY 8(f)
next #1) run WHERE THE MODEL RUNS -- template OBJ -> get_model -> A / D / U on the device -> one HIP train step --
held to the reference-generated fixture of the same template (tests/golden/hier_torus5k.npz) and to the CPU
oracle on that fixture's topology.  (The CPU tests of the generator only ever run in the build container.)"""
import os
import time

import numpy as np
import pytest
import torch

from conftest import CFG_5K, load_golden
from meshgen import torus_mesh

pytestmark = pytest.mark.gpu


def test_template_to_train_step_on_the_gpu_box(tmp_path, capsys):
    from model import get_model
    from oracle import cheb_oracle as O
    dev = torch.device("cuda:0")
    topo = load_golden("hier_torus5k.npz")
    v, f = torus_mesh(51, 98)                    # 4998 vertices / 9996 faces / genus 1: the template's counts
    obj = tmp_path / "torus5k.obj"
    with open(obj, "w") as fp:
        for p in v:
            fp.write("v %.17g %.17g %.17g\n" % tuple(p))
        for a, b, c in f:
            fp.write(f"f {a + 1} {b + 1} {c + 1}\n")
    cfg = dict(CFG_5K, dropout=0.0, template=str(obj), downsampling_factors=[4, 4, 4, 4], type="cheb_VAE",
               model="optimal_sigma_VAE", checkpoint_dir=str(tmp_path))
    torch.manual_seed(666)
    t0 = time.time()
    net = get_model(cfg, dev)
    build_s = time.time() - t0
    capsys.readouterr()
    assert net.num_nodes == [int(x) for x in topo["num_nodes"]] == [4998, 1250, 313, 79, 20]
    # the generated operators are the reference generator's: adjacency and decimation entry for entry, upsampling
    # weights to 1e-6 (the fixture's closest-point search is the stand-in's)
    for i, a in enumerate(net.adjacency_matrices):
        idx = a._indices().cpu().numpy()
        assert np.array_equal(idx[0], topo[f"A{i}_row"]) and np.array_equal(idx[1], topo[f"A{i}_col"]), f"A{i}"
    for i, d in enumerate(net.downsample_matrices):
        idx = d._indices().cpu().numpy()
        assert np.array_equal(idx[0], topo[f"D{i}_row"]) and np.array_equal(idx[1], topo[f"D{i}_col"]), f"D{i}"
    for i, u in enumerate(net.upsample_matrices):
        idx = u._indices().cpu().numpy()
        assert np.array_equal(idx[0], topo[f"U{i}_row"]) and np.array_equal(idx[1], topo[f"U{i}_col"]), f"U{i}"
        np.testing.assert_allclose(u._values().cpu().numpy(), topo[f"U{i}_val"], rtol=0, atol=1e-6)

    # one train-mode forward + backward of the model built on the GENERATED hierarchy against the oracle built on
    # the FIXTURE's hierarchy (same weights): forward 1e-4, gradients 1e-4 relative
    B = 3
    x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(0))
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2)

    class Data:
        pass

    d = Data()
    d.x, d.num_graphs, d.edge_index = x.to(dev).reshape(-1, 3), B, None
    net.train()
    torch.manual_seed(123)
    loss, correct, recon, (kld, rec, z), y_hat = net(d, x.to(dev), y.to(dev), m_type="train")
    loss.backward()
    torch.cuda.synchronize()
    ora = O.OracleVAE(cfg, O.Topology(topo), {k: t.cpu() for k, t in net.state_dict().items()}, requires_grad=True)
    ora.training = True
    torch.manual_seed(123)
    lo, co, ro, (ko, reco, zo), yo, _, _ = ora.forward(x, x.clone(), y, "train")
    lo.backward()
    assert (recon.detach().cpu() - ro.detach()).abs().max().item() < 1e-4
    assert abs(loss.item() - lo.item()) < 1e-4 * max(1.0, abs(lo.item()))
    for k, g in ora.grads().items():
        got = dict(net.named_parameters())[k].grad.cpu()
        rel = (got - g).norm().item() / max(g.norm().item(), 1e-12)
        assert rel < 1e-4, (k, rel)
    print(f"hierarchy of the 4998-vertex torus built on this host in {build_s:.2f} s (incl. model construction)")
