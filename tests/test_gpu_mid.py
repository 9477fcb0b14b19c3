"""GPU (-m gpu): the small-level decoder sub-network as ONE launch (csrc/cheb_mid.hip: unpool, ChebConv + ReLU, unpool,
ChebConv + ReLU, unpool; cheb_VAE.py:281-286 with nn/conv.py:557-577 and nn/pool.py:17-20) against the per-layer
kernels it replaces (debug switch no_mid), on the 5k model.  The reference vectors of the whole model
(test_gpu_parity.py, test_gpu_bf16.py) run through the fused launch too; this file pins the DIFFERENCE between the
two kernel families: same algorithm, direct Chebyshev evaluation instead of Clenshaw and a different fp32 summation
order, so outputs agree to a few ulp of the largest intermediate (bar 2e-5 relative to max|.|, gradients 1e-4 of the
tensor norm: a ReLU sign can flip on a pre-activation within 1e-7 of zero), not bitwise.
"""
import os

import pytest
import torch

from conftest import CFG_5K, ROOT

pytestmark = pytest.mark.gpu


def _model(dev):
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
    torch.manual_seed(666)
    return cheb_VAE(3, dict(CFG_5K, dropout=0.0), D, U, A, nn_, model="optimal_sigma_VAE").to(dev)


@pytest.mark.parametrize("storage,bar,gbar", [("f32", 2e-5, 1e-4), ("bf16", 1e-2, 0.2)])
@pytest.mark.parametrize("B", [1, 5])
def test_fused_decoder_head_matches_per_layer_kernels(storage, bar, gbar, B):
    from meshvae_hip import debug_switch, lib
    from meshvae_hip.engine import NativeStep
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, CFG_5K["num_style"], generator=torch.Generator().manual_seed(6)).to(dev)
    res = []
    for no_mid in (1, 0):
        net = _model(dev).train()
        nat = NativeStep(net, B, storage=storage)
        with debug_switch("no_mid", no_mid):
            assert lib().mvh_debug_get(b"no_mid") == no_mid
            loss, corr, recon, (kld, rec, z), yh = nat.forward_backward(x, x, y, eps=eps)
            torch.cuda.synchronize()
        res.append(dict(loss=float(loss), recon=recon.clone(), z=z.clone(),
                        grads={k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    ref, got = res
    assert torch.equal(ref["z"], got["z"])                       # the encoder is untouched
    scale = float(ref["recon"].abs().max())
    e = float((ref["recon"] - got["recon"]).abs().max()) / scale
    assert e < bar, e
    assert abs(ref["loss"] - got["loss"]) <= (1e-6 if storage == "f32" else 1e-4) * abs(ref["loss"])
    worst = 0.0
    for k, g in ref["grads"].items():
        rel = float((g - got["grads"][k]).norm()) / max(float(g.norm()), 1e-12)
        worst = max(worst, rel)
        assert rel < gbar, (k, rel)
    print(f"[mid {storage} B={B}] recon {e:.2e} of max|recon|, worst gradient rel {worst:.2e}")
    if storage == "f32":
        assert not torch.equal(ref["recon"], got["recon"]) or e == 0.0
