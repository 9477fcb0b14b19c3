"""GPU: the train-step engine -- fused Adam vs torch.optim.Adam, flat-buffer gradients vs plain
autograd, hipGraph replay vs eager, and a short training run that must reduce the loss."""
import os

import pytest
import torch

from conftest import ROOT, TINY_CFG

pytestmark = pytest.mark.gpu


def _net(dev, dropout=0.0, seed=666):
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), dev)
    torch.manual_seed(seed)
    return cheb_VAE(3, dict(TINY_CFG, dropout=dropout), D, U, A, nn_).to(dev)


def test_fused_adam_matches_torch_adam():
    from meshvae_hip.engine import FlatParams, FusedAdam
    dev = torch.device("cuda:0")
    lin = torch.nn.Linear(37, 19).to(dev)
    ref = torch.nn.Linear(37, 19).to(dev)
    ref.load_state_dict(lin.state_dict())
    flat = FlatParams(lin)
    opt = FusedAdam(flat, lr=1e-3, weight_decay=5e-4)
    topt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=5e-4)
    g = torch.Generator(device="cpu").manual_seed(3)
    for _ in range(5):
        grads = [torch.randn(p.shape, generator=g).to(dev) for p in ref.parameters()]
        flat.zero_grad()
        for p, rp, gr in zip(lin.parameters(), ref.parameters(), grads):
            p.grad += gr
            rp.grad = gr.clone()
        opt.step()
        topt.step()
    for p, rp in zip(lin.parameters(), ref.parameters()):
        torch.testing.assert_close(p, rp, rtol=1e-5, atol=1e-7)
    assert int(opt.step_count) == 5


def test_fused_adam_vector_and_scalar_paths_agree_bitwise():
    """k_adam updates four parameters per thread through 16-byte accesses when all four buffers are 16-byte aligned and
    the group is not cut by the no-gradient range; element-wise otherwise.  Both forms are the same arithmetic: buffers
    shifted by one float (scalar path everywhere) give bitwise the aligned run's result, for a size that is not a
    multiple of four, a skip range that starts and ends inside groups, the host-counted and the device-counted form."""
    from meshvae_hip import check, lib
    dev = torch.device("cuda:0")
    n, lo, hi = 4099, 1022, 1031
    g = torch.Generator().manual_seed(5)
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g)
    st = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for shift in (0, 1):
        for counted in (True, False):
            bufs = [torch.zeros(n + 4, device=dev) for _ in range(4)]
            p, gr, m, v = (b[shift:shift + n] for b in bufs)
            p.copy_(p0)
            gr.copy_(g0)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            for step in (1, 2, 3):
                args = (p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 5e-4, 0.5, cnt.data_ptr())
                if counted:
                    check(lib().mvh_adam_step_counted(st, *args, step, lo, hi))
                else:
                    check(lib().mvh_adam_step(st, *args, lo, hi))
            torch.cuda.synchronize()
            assert int(cnt) == 3
            assert all(float(b[:shift].abs().sum()) == 0 and float(b[shift + n:].abs().sum()) == 0 for b in bufs[2:])   # no stray writes
            res[(shift, counted)] = (p.cpu().clone(), m.cpu().clone(), v.cpu().clone())
    ref = res[(0, True)]
    assert torch.equal(ref[0][lo:hi], p0[lo:hi]) and float(ref[1][lo:hi].abs().sum()) == 0     # the skipped range is untouched
    assert not torch.equal(ref[0][:lo], p0[:lo])
    for key, got in res.items():
        for a, b in zip(got, ref):
            assert torch.equal(a, b), key
    # ... and against torch.optim.Adam on the same numbers (grad_scale folded into the gradient)
    tp = torch.nn.Parameter(p0.clone().to(dev))
    topt = torch.optim.Adam([tp], lr=1e-3, weight_decay=5e-4)
    for _ in range(3):
        tp.grad = (g0 * 0.5).to(dev)
        topt.step()
    keep = torch.ones(n, dtype=torch.bool)
    keep[lo:hi] = False
    torch.testing.assert_close(ref[0][keep], tp.detach().cpu()[keep], rtol=1e-5, atol=1e-7)


def test_trainstep_graph_equals_eager_and_learns():
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    B = 8
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(0))
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2)
    losses = {}
    params = {}
    for mode in ("eager", "graph"):
        net = _net(dev, dropout=0.0)
        net.train()
        step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=(mode == "graph"))
        step.load(x.to(dev), x.to(dev), y.to(dev))
        init = step.flat.param.clone()
        if step.use_graph:
            step.capture(warmup=1)                 # warm-up steps touch params/optimizer state: undo
        step.flat.param.copy_(init)
        step.opt.exp_avg.zero_(), step.opt.exp_avg_sq.zero_(), step.opt.step_count.zero_()
        torch.manual_seed(7)                       # host-side eps stream identical in both modes
        ls = []
        for _ in range(6):
            loss, correct, recon = step.step()
            ls.append(float(loss))
        losses[mode] = ls
        params[mode] = step.flat.param.clone()
    assert losses["eager"][-1] < losses["eager"][0]                 # Adam makes progress
    torch.testing.assert_close(torch.tensor(losses["graph"]), torch.tensor(losses["eager"]), rtol=1e-6, atol=1e-3)
    torch.testing.assert_close(params["graph"], params["eager"], rtol=1e-5, atol=1e-6)


def test_trainstep_micro_batched_chains_equal_one_chain():
    """n_micro = 2: the per-GPU batch as two independent chains, each enqueued by its own host thread on
    its own stream, gradients averaged before Adam -- the same update as one chain (the loss is a mean over
    meshes), up to fp32 summation order."""
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    B = 8
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(0))
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2)
    params, losses = {}, {}
    for n_micro in (1, 2):
        net = _net(dev, dropout=0.0)
        net.train()
        step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=False, n_micro=n_micro)
        assert step.n_micro == n_micro
        step.load(x.to(dev), x.to(dev), y.to(dev))
        torch.manual_seed(7)
        for _ in range(4):
            loss, correct, recon = step.step()
        torch.cuda.synchronize()
        params[n_micro], losses[n_micro] = step.flat.param.clone(), float(loss)
        assert recon.shape == (B, 162, 3)
    assert abs(losses[1] - losses[2]) <= 1e-5 * abs(losses[1]) + 1e-3
    torch.testing.assert_close(params[2], params[1], rtol=1e-4, atol=2e-6)


def test_trainstep_graph_two_chains():
    """The hipGraph topology the engine supports for micro-batched chains (n_micro = 2, use_graph = True): chain 1 is
    one fork of the capture stream and keeps its weight-gradient kernels inline (a second-level fork kills
    hipStreamEndCapture on ROCm 7.2, tools/graph_diag.py).  Replay must equal the eager single chain."""
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    B = 8
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    res = {}
    for mode, n_micro, graph in (("eager1", 1, False), ("graph2", 2, True)):
        net = _net(dev, dropout=0.0).train()
        step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=graph, n_micro=n_micro)
        assert step.n_micro == n_micro
        step.load(x, x, y)
        init = step.flat.param.clone()
        if graph:
            step.capture(warmup=1)
        step.flat.param.copy_(init)
        step.opt.exp_avg.zero_(), step.opt.exp_avg_sq.zero_(), step.opt.step_count.zero_()
        torch.manual_seed(7)
        for _ in range(3):
            step.step()
        torch.cuda.synchronize()
        res[mode] = (step.flat.param.clone(), float(step.out[0]))
    torch.testing.assert_close(res["graph2"][0], res["eager1"][0], rtol=1e-4, atol=2e-6)
    assert abs(res["graph2"][1] - res["eager1"][1]) <= 1e-5 * abs(res["eager1"][1]) + 1e-3


def test_two_rank_emulation_equals_one_large_batch():
    """Data-parallel arithmetic without a second GPU: "rank 0" and "rank 1" each run forward+backward on their
    shard of a global batch (engine.shard_range), the flat gradients are SUMMED (what the all-reduce does) and the
    fused Adam applies them with grad_scale = 1/world -- the result must be the single-rank step on the whole batch
    (the loss is a per-rank mean over meshes, cheb_VAE.py:342, and the shards are equal)."""
    from meshvae_hip.engine import TrainStep, shard_range
    dev = torch.device("cuda:0")
    G, world = 8, 2
    x = torch.randn(G, 162, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(G) % 2, 2).to(dev)
    eps = torch.randn(G, 16, generator=torch.Generator().manual_seed(5)).to(dev)
    net1 = _net(dev, dropout=0.0).train()
    big = TrainStep(net1, G, lr=1e-3, weight_decay=5e-4, use_graph=False)
    big.load(x, x, y)
    big._draw_eps = lambda: big.eps.copy_(eps)
    big.step()
    nets = [_net(dev, dropout=0.0).train() for _ in range(world)]
    ranks = []
    for r, net in enumerate(nets):
        lo, hi = shard_range(G, r, world)
        st = TrainStep(net, hi - lo, lr=1e-3, weight_decay=5e-4, use_graph=False)
        st.load(x[lo:hi], x[lo:hi], y[lo:hi])
        st.eps.copy_(eps[lo:hi])
        st._fwd_bwd()                                  # the rank's local backward
        ranks.append(st)
    total = ranks[0].flat.grad + ranks[1].flat.grad    # sum all-reduce
    for st in ranks:
        st.flat.grad.copy_(total)
        st.opt.step(1.0 / world)                       # the 1/world scale lives in the Adam kernel
    torch.cuda.synchronize()
    assert torch.equal(ranks[0].flat.param, ranks[1].flat.param)
    torch.testing.assert_close(ranks[0].flat.param, big.flat.param, rtol=1e-4, atol=2e-6)


def test_trainstep_leaves_unused_dec_lin_1_alone():
    """torch.optim.Adam never touches dec_lin_1 (no gradient: cheb_VAE.py:165); the fused Adam over the flat buffer
    must not decay it either (coupled weight decay on a zero gradient would shrink it every step)."""
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    net = _net(dev, dropout=0.2).train()
    w0, b0 = net.dec_lin_1.weight.detach().clone(), net.dec_lin_1.bias.detach().clone()
    other0 = net.dec_lin_2.weight.detach().clone()
    step = TrainStep(net, 4, lr=1e-2, weight_decay=5e-2, use_graph=False)
    x = torch.randn(4, 162, 3, device=dev)
    step.load(x, x, torch.nn.functional.one_hot(torch.arange(4) % 2, 2).to(dev))
    for _ in range(5):
        step.step()
    torch.cuda.synchronize()
    assert torch.equal(net.dec_lin_1.weight, w0) and torch.equal(net.dec_lin_1.bias, b0)
    assert not torch.equal(net.dec_lin_2.weight, other0)
    lo, hi = step.opt.skip
    assert float(step.opt.exp_avg[lo:hi].abs().sum()) == 0.0 and float(step.opt.exp_avg_sq[lo:hi].abs().sum()) == 0.0


def test_eps_provider_only_serves_its_own_batch_size():
    """TrainStep's static noise buffer is handed to the module only for the batch size it was built for; another
    batch size through net(...) draws fresh host noise like the reference (no out-of-bounds read of the buffer)."""
    from meshvae_hip.engine import TrainStep, _Batch
    dev = torch.device("cuda:0")
    net = _net(dev, dropout=0.0).train()
    step = TrainStep(net, 4, use_graph=False)
    assert net._eps_provider(4, net.z, dev) is step.eps and net._eps_provider(6, net.z, dev) is None
    x = torch.randn(6, 162, 3, device=dev)
    y = torch.nn.functional.one_hot(torch.arange(6) % 2, 2).to(dev)
    torch.manual_seed(11)
    z1 = net(_Batch(x), x, y, m_type="train")[3][2].clone()
    torch.manual_seed(11)
    z2 = net(_Batch(x), x, y, m_type="train")[3][2].clone()
    torch.manual_seed(12)
    z3 = net(_Batch(x), x, y, m_type="train")[3][2].clone()
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)      # noise follows the host generator, B = 6 rows of it


def test_module_call_after_draw_ahead_step_samples_noise():
    """with a private host generator the eager native step takes its noise from a draw-ahead block and never
    writes TrainStep.eps -- a module-level train-mode call of the same batch size must then draw its own host noise (as the
    reference does, cheb_VAE.py:316), not read that buffer's zeros: z != mu, and it follows the process generator."""
    from meshvae_hip.engine import TrainStep, _Batch
    dev = torch.device("cuda:0")
    net = _net(dev, dropout=0.0).train()
    step = TrainStep(net, 4, use_graph=False, noise_seed=5)
    assert TrainStep.__doc__ and step._eps_ahead()
    x = torch.randn(4, 162, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(4) % 2, 2).to(dev)
    step.load(x, x, y)
    step.step()
    assert net._eps_provider(4, net.z, dev) is None and float(step.eps.abs().sum()) == 0.0
    with torch.no_grad():
        net.eval()
        mu = net(_Batch(x), x, y, m_type="test")[3][2].clone()          # test mode: z_ = mu
        net.train()
        torch.manual_seed(21)
        z1 = net(_Batch(x), x, y, m_type="train")[3][2].clone()
        torch.manual_seed(21)
        z2 = net(_Batch(x), x, y, m_type="train")[3][2].clone()
    assert torch.equal(z1, z2) and not torch.equal(z1, mu)
    assert float((z1 - mu).abs().mean()) > 0.1                           # eps * std with std ~ 1 at initialisation


def test_trainstep_private_noise_generators():
    """noise_seed gives the step generators of its own (seed + rank): two steps with the same seed replay the same
    loss sequence whatever the process-wide generators did in between; different seeds differ."""
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    x = torch.randn(4, 162, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(4) % 2, 2).to(dev)
    runs = []
    for seed, junk in ((5, 0), (5, 3), (6, 0)):
        net = _net(dev, dropout=0.2).train()
        step = TrainStep(net, 4, use_graph=False, noise_seed=seed)
        step.load(x, x, y)
        for _ in range(junk):
            torch.randn(3), torch.rand(3, device=dev)          # disturb the default generators
        runs.append([float(step.step()[0]) for _ in range(3)])
    assert runs[0] == runs[1] and runs[0] != runs[2]


def test_trainstep_graph_with_private_noise_replays_fresh_masks():
    """hipGraph replay with the step's private generators (noise_seed): the dropout uniforms are drawn OUTSIDE the
    graph into static buffers before every replay, so two replays use different masks (a generator consumed inside the
    capture would bake one offset into the graph) and the loss sequence equals the eager run with the same seed."""
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    B = 4
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    seqs, masks = {}, {}
    for mode in ("eager", "graph"):
        net = _net(dev, dropout=0.2).train()
        step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=(mode == "graph"), noise_seed=3)
        step.load(x, x, y)
        init = step.flat.param.clone()
        if step.use_graph:
            step.capture(warmup=2)                 # consumes noise and touches params / Adam state: reset both
        from meshvae_hip.engine import rank_generators
        step.host_gen, step.dev_gen = rank_generators(3, 0, dev)
        step._u_left = 0                           # (drop the uniforms drawn ahead with the old generator)
        step.flat.param.copy_(init)
        step.opt.exp_avg.zero_(), step.opt.exp_avg_sq.zero_(), step.opt.step_count.zero_()
        step.opt._host_step = None
        ls, us = [], []
        for _ in range(4):
            ls.append(float(step.step()[0]))
            us.append(step._u_bufs[0].clone())
        seqs[mode], masks[mode] = ls, us
    assert not torch.equal(masks["graph"][0], masks["graph"][1])            # every replay has its own uniforms
    assert all(torch.equal(a, b) for a, b in zip(masks["graph"], masks["eager"]))
    torch.testing.assert_close(torch.tensor(seqs["graph"]), torch.tensor(seqs["eager"]), rtol=1e-6, atol=1e-3)
    assert len(set(seqs["graph"])) == 4


def test_flat_grads_equal_plain_autograd():
    from meshvae_hip.engine import FlatParams
    dev = torch.device("cuda:0")
    B = 4
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)

    class D:
        pass

    d = D()
    d.x, d.num_graphs, d.edge_index = x.reshape(-1, 3), B, None
    a, b = _net(dev), _net(dev)
    a.eval(), b.eval()
    a(d, x, y)[0].backward()
    flat = FlatParams(b)
    flat.zero_grad()
    b(d, x, y)[0].backward()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert float(pb.grad.abs().sum()) == 0.0, k          # dec_lin_1: zero-filled, no special casing
        else:
            assert torch.equal(pa.grad, pb.grad), k


@pytest.mark.parametrize("gt_dtype", [torch.float32, torch.float64])
def test_native_step_equals_module_path(gt_dtype):
    """mvh_vae_forward/backward (one C++ launch sequence, dW on a side stream) must reproduce the
    per-module autograd path: same kernels, so outputs and gradients agree to the last bit."""
    from meshvae_hip.engine import NativeStep
    dev = torch.device("cuda:0")
    B = 6
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, 16, generator=torch.Generator().manual_seed(3)).to(dev)

    class D:
        pass

    d = D()
    d.x, d.num_graphs, d.edge_index = x.reshape(-1, 3), B, None
    a, b = _net(dev), _net(dev)
    a.fused_step = False                                      # the per-module autograd path
    a.train(), b.train()                                      # dropout p = 0: deterministic
    a._eps_provider = lambda B_, Z_, dev_: eps
    loss_a, corr_a, recon_a, (kld_a, rec_a, z_a), yh_a = a(d, x.to(gt_dtype), y, m_type="train")
    loss_a.backward()
    nat = NativeStep(b, B)
    loss_b, corr_b, recon_b, (kld_b, rec_b, z_b), yh_b = nat.forward_backward(x, x.to(gt_dtype), y, eps=eps)
    torch.cuda.synchronize()
    assert loss_b.dtype == gt_dtype and rec_b.dtype == gt_dtype
    for u, v in ((loss_a, loss_b), (recon_a, recon_b), (kld_a, kld_b), (rec_a, rec_b), (z_a, z_b), (yh_a, yh_b)):
        assert torch.equal(u.detach(), v), "forward differs"
    assert int(corr_a) == int(corr_b)
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert float(pb.grad.abs().sum()) == 0.0, k
        else:
            assert torch.equal(pa.grad, pb.grad), k
    # eval path (no eps): z = mu
    a.eval(), b.eval()
    with torch.no_grad():
        la = a(d, x.to(gt_dtype), y, m_type="test")
    lb = nat.forward_backward(x, x.to(gt_dtype), y, eps=None, backward=False)
    assert torch.equal(la[0], lb[0]) and torch.equal(la[2], lb[2]) and torch.equal(la[3][2], lb[3][2])


@pytest.mark.parametrize("cfg_over, B", [
    ({"polygon_order": [1, 2, 3]}, 5),                       # K = 1 and 2 layers, batch not a multiple of 8
    ({"polygon_order": [6, 4, 6]}, 3),
    ({"num_conv_filters": [5, 7, 7]}, 2),                    # channel counts outside the LDS kernels: general pipeline
])
def test_native_step_unusual_configs_match_module_path(cfg_over, B):
    """Engine vs per-module autograd path on configurations the 5k default never exercises."""
    from meshvae_hip.engine import NativeStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), dev)
    cfg = dict(TINY_CFG, dropout=0.0, **cfg_over)
    nets = []
    for _ in range(2):
        torch.manual_seed(11)
        nets.append(cheb_VAE(3, cfg, D, U, A, nn_).to(dev).train())
    a, b = nets
    a.fused_step = False                                      # the per-module autograd path
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, 16, generator=torch.Generator().manual_seed(5)).to(dev)

    class Dt:
        pass
    d = Dt()
    d.x, d.num_graphs, d.edge_index = x.reshape(-1, 3), B, None
    a._eps_provider = lambda B_, Z_, dev_: eps
    loss_a, _, recon_a, _, _ = a(d, x.double(), y, m_type="train")
    loss_a.backward()
    loss_b, _, recon_b, _, _ = NativeStep(b, B).forward_backward(x, x.double(), y, eps=eps)
    torch.cuda.synchronize()
    torch.testing.assert_close(recon_b, recon_a.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss_b, loss_a.detach(), rtol=1e-9, atol=1e-6)
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert float(pb.grad.abs().sum()) == 0.0, k
        else:
            err = float((pa.grad - pb.grad).norm()) / max(float(pa.grad.norm()), 1e-12)
            assert err < 1e-4, (k, err)


@pytest.mark.parametrize("which,B,dropout", [("tiny", 5, 0.2), ("5k", 3, 0.2), ("5k", 2, 0.0)])
def test_fused_head_kernels_are_bitwise_the_separate_launches(which, B, dropout):
    """dec_lin (cheb_VAE.py:277) rides inside the latent-head launches of the native step, forward and dX (dense.hip:
    k_latent_fwd / k_latent_bwd evaluate it with the matrix instructions, lanes and summation order of k_gemm16).
    Against the same step with the debug switch no_head_fuse (dec_lin as its own mvh_linear_fwd / mvh_linear_bwd GEMM)
    every output and every gradient must be IDENTICAL, with and without dropout, for H = 64 (4 K-waves) and H = 512 (8)."""
    from conftest import CFG_5K
    from meshvae_hip import debug_switch
    from meshvae_hip.engine import NativeStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    cfg, topo, N = (TINY_CFG, "topology_tiny.npz", 162) if which == "tiny" else (CFG_5K, "topology_5k.npz", 4998)
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", topo), dev)
    x = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(21)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, 16, generator=torch.Generator().manual_seed(22)).to(dev)
    res = []
    for no_fuse in (1, 0):
        torch.manual_seed(5)
        net = cheb_VAE(3, dict(cfg, dropout=dropout), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
        nat = NativeStep(net, B)
        drop_u = None
        if dropout > 0:                                          # the dropout uniforms of this step [B, 3 H + flat]
            drop_u = torch.rand(B, nat.u_cols, generator=torch.Generator().manual_seed(77)).to(dev)
        with debug_switch("no_head_fuse", no_fuse):
            loss, corr, recon, (kld, rec, z), yh = nat.forward_backward(x, x, y, eps=eps, drop_u=drop_u)
            torch.cuda.synchronize()
        res.append((loss.clone(), recon.clone(), z.clone(), yh.clone(),
                    {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    a, b = res
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u, v)
    assert sorted(a[4]) == sorted(b[4])
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k


def test_dense_gradients_are_final_at_the_library_event():
    """The overlapped all-reduce (engine.TrainStep._all_reduce_overlapped) reads the dense-layer gradients on
    another stream as soon as the event behind mvh_vae_wait_dense_grads fires, while the encoder half of the
    backward is still running.  Poison the gradient buffer, snapshot the dense tail on a second stream behind
    that event, and require the snapshot to equal the final gradients (5k model, so the backward is long)."""
    import threading

    import meshvae_hip
    from conftest import CFG_5K
    from meshvae_hip.engine import TrainStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
    torch.manual_seed(666)
    net = cheb_VAE(3, CFG_5K, D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
    B = 16
    step = TrainStep(net, B, use_graph=False)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, nn_[0], 3, generator=g)
    step.load(x, x, torch.nn.functional.one_hot(torch.arange(B) % 2, 2))
    flat = step.flat
    split = flat.conv_dense_split()
    k = flat.offsets.index(split)
    assert flat.names[k] == "classifier_layer.weight" and all(n.startswith("cheb") for n in flat.names[:k])
    assert (flat.numel - split) * 4 > 0.95 * flat.numel * 4          # the tail is almost all of the bytes
    comm = torch.cuda.Stream(dev)
    snap = torch.empty_like(flat.grad[split:])
    for _ in range(3):
        flat.grad.fill_(float("nan"))
        step._draw_noise()
        step._fwd_bwd()
        meshvae_hip.check(meshvae_hip.lib().mvh_vae_wait_dense_grads(comm.cuda_stream))
        with torch.cuda.stream(comm):
            snap.copy_(flat.grad[split:])
        torch.cuda.synchronize()
        for name, p, off in zip(flat.names[k:], flat.params[k:], flat.offsets[k:]):
            got = snap[off - split:off - split + p.numel()]
            if name.startswith("dec_lin_1."):      # no gradient exists for it (cheb_VAE.py:165): the span is never written
                assert not torch.isfinite(got).any()
                continue
            assert torch.isfinite(got).all(), "dense gradient read before it was written"
            assert torch.equal(got, p.grad.reshape(-1))
        for p in flat.params[:k]:
            assert torch.isfinite(p.grad).all()
    flat.grad.zero_()
    # the gradient lanes (and this event) belong to the DEVICE, not to the host thread that issued the backward (round 4:
    # a second thread's lanes of its own were what made the autograd-driven module path slow): any thread may wait
    err = []

    def other():
        try:
            with torch.cuda.device(dev):
                meshvae_hip.check(meshvae_hip.lib().mvh_vae_wait_dense_grads(comm.cuda_stream))
        except meshvae_hip.MeshVaeHipError as e:
            err.append(str(e))
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert not err, err


def test_native_step_hires_20k_equals_module_path():
    """BASELINE configs[3] through the native step: level 0 (19 992 vertices) runs the stack pipeline, whose
    T_k stacks the native forward keeps for the backward (the module path rebuilds them) and whose weight
    gradients / G stacks run on the matrix pipe -- same kernels either way, so the two paths must agree."""
    from conftest import CFG_20K
    from meshvae_hip.engine import NativeStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_20k.npz"), dev)
    B = 3
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, nn_[0], 3, generator=g).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, 16, generator=g).to(dev)

    class D_:
        pass

    d = D_()
    d.x, d.num_graphs, d.edge_index = x.reshape(-1, 3), B, None
    nets = []
    for _ in range(2):
        torch.manual_seed(666)
        nets.append(cheb_VAE(3, dict(CFG_20K, dropout=0.0), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train())
    a, b = nets
    a.fused_step = False                                      # the per-module autograd path
    a._eps_provider = lambda B_, Z_, dev_: eps
    loss_a, _, recon_a, _, _ = a(d, x.double(), y, m_type="train")
    loss_a.backward()
    loss_b, _, recon_b, _, _ = NativeStep(b, B).forward_backward(x, x.double(), y, eps=eps)
    torch.cuda.synchronize()
    assert torch.equal(recon_a.detach(), recon_b)
    torch.testing.assert_close(loss_b, loss_a.detach(), rtol=1e-12, atol=0)
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert float(pb.grad.abs().sum()) == 0.0, k
        else:
            err = float((pa.grad - pb.grad).norm()) / max(float(pa.grad.norm()), 1e-20)
            assert err < 1e-6, (k, err)


def test_hires_20k_first_layer_stack_at_the_pooled_rows_only():
    """configs[3], first layer: k_cheb_big keeps T_k x at the rows the pooling selects (the stack layout of cheb_tstack.hip),
    k_stack_contract forms the pooled output from it and k_stack_dw the weight gradient -- against the full-stack pipeline
    (debug switch no_big_tstack): outputs bitwise (the contraction keeps k_cheb_contract's fma order), gradients to
    fp32 reassociation (1e-6; the first layer's own weight gradient is summed in another order), and the switch matters."""
    from conftest import CFG_20K
    from meshvae_hip import debug_switch
    from meshvae_hip.engine import NativeStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_20k.npz"), dev)
    B = 2
    g = torch.Generator().manual_seed(18)
    x = torch.randn(B, nn_[0], 3, generator=g).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, 16, generator=g).to(dev)
    res = []
    for sw in (0, 1):
        torch.manual_seed(666)
        net = cheb_VAE(3, dict(CFG_20K, dropout=0.0), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
        with debug_switch("no_big_tstack", sw):
            loss, _, recon, _, _ = NativeStep(net, B).forward_backward(x, x.double(), y, eps=eps)
            torch.cuda.synchronize()
        res.append((loss.clone(), recon.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0])
    differs = False
    for k, ga in res[0][2].items():
        gb = res[1][2][k]
        err = float((ga - gb).norm()) / max(float(gb.norm()), 1e-20)
        assert err < 1e-6, (k, err)
        differs = differs or not torch.equal(ga, gb)
    assert differs, "no_big_tstack changed nothing: the selected-rows stack path did not run"


def test_module_forward_is_one_fused_autograd_node():
    """cheb_VAE.forward under grad mode runs the native step behind a single autograd node (what main.py's
    `loss.backward()` then triggers): same numbers as the per-module path, gradients ACCUMULATE like any autograd
    result, dec_lin_1 keeps grad None, and a backward through stale activations is refused."""
    dev = torch.device("cuda:0")
    B = 5
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 162, 3, generator=g).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, 16, generator=g).to(dev)

    class D:
        pass

    d = D()
    d.x, d.num_graphs, d.edge_index = x.reshape(-1, 3), B, None
    a, b = _net(dev), _net(dev)
    a.fused_step = False
    for n_ in (a, b):
        n_.train()
        n_._eps_provider = lambda B_, Z_, dev_: eps
    out_a = a(d, x.double(), y, m_type="train")
    out_b = b(d, x.double(), y, m_type="train")
    assert out_b[0].grad_fn is not None and type(out_b[0].grad_fn).__name__.startswith("_FusedModelFn")
    assert not out_b[2].requires_grad and out_b[0].dtype == torch.float64
    for u, v in zip((out_a[0], out_a[2], out_a[3][0], out_a[3][1], out_a[3][2], out_a[4]),
                    (out_b[0], out_b[2], out_b[3][0], out_b[3][1], out_b[3][2], out_b[4])):
        assert torch.equal(u.detach(), v.detach())
    assert int(out_a[1]) == int(out_b[1])
    (out_a[0] * 0.5).backward()
    (out_b[0] * 0.5).backward()                               # an upstream factor reaches every gradient
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None, k                         # dec_lin_1
        else:
            torch.testing.assert_close(pb.grad, pa.grad, rtol=1e-6, atol=1e-7, msg=k)
    first = {k: p.grad.clone() for k, p in b.named_parameters() if p.grad is not None}
    b(d, x.double(), y, m_type="train")[0].backward()         # no zero_grad in between: 0.5 g + g
    for k, p in b.named_parameters():
        if p.grad is not None:
            torch.testing.assert_close(p.grad, 3.0 * first[k], rtol=1e-5, atol=1e-6, msg=k)
    stale = b(d, x.double(), y, m_type="train")[0]
    b(d, x.double(), y, m_type="train")
    with pytest.raises(RuntimeError, match="must follow its own forward"):
        stale.backward()
    # no-grad evaluation (main.py:129): the same native launch sequence, forward only, equal to the per-module path
    a.eval(), b.eval()
    with torch.no_grad():
        ea, eb = a(d, x, y, m_type="test"), b(d, x, y, m_type="test")
    assert eb[0].grad_fn is None
    for u, v in zip((ea[0], ea[2], ea[3][0], ea[3][1], ea[3][2], ea[4]), (eb[0], eb[2], eb[3][0], eb[3][1], eb[3][2], eb[4])):
        assert torch.equal(u, v)
    assert int(ea[1]) == int(eb[1])


def test_piecewise_inference_calls_take_the_native_sequences():
    """net.encoder / net.sample under no_grad (inference.py, crecon.py) run mvh_vae_encode / mvh_vae_decode: the
    same kernels as the per-module path, so the results are identical -- in eval mode and, with the same torch seed,
    with the dropout the reference leaves on in crecon.py."""
    dev = torch.device("cuda:0")
    B = 6
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 162, 3, generator=g).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    a, b = _net(dev, dropout=0.3), _net(dev, dropout=0.3)
    a.fused_step = False
    for mode in ("eval", "train"):
        for n_ in (a, b):
            n_.train(mode == "train")
        outs = []
        for n_ in (a, b):
            torch.manual_seed(77)
            with torch.no_grad():
                h = n_.encoder(x)
                y_hat = n_.classifier(h)
                mu = n_.z_mean(torch.cat([y.float(), h], -1))
                rec = n_.sample(y, mu)
                rec2 = n_.sample(1 - y, mu)
            outs.append((h, y_hat, rec, rec2))
        for u, v in zip(*outs):
            assert torch.equal(u, v), mode
        assert outs[1][2].shape == (B, 162, 3)
    # under grad mode the piecewise calls stay on the differentiable per-module path
    b.train()
    h = b.encoder(x)
    assert h.requires_grad


def test_piecewise_inference_on_the_5k_model_runs_the_patch_kernels():
    """... on the 5k template the native encode / decode sequences run the vertex-patch kernels (first layer:
    k_patch_enc0, last decoder stage: k_patch_fwd) where the per-module path runs the slab kernels: the same numbers to
    fp32 reassociation (and not bitwise -- the patch kernels really ran), also with the patch kernels switched off."""
    from meshvae_hip import debug_switch
    dev = torch.device("cuda:0")
    B = 5
    x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(10)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    a, b = _ref_model("5k", dev).eval(), _ref_model("5k", dev).eval()
    a.fused_step = False

    def run(n_):
        with torch.no_grad():
            h = n_.encoder(x)
            mu = n_.z_mean(torch.cat([y.float(), h], -1))
            return h, n_.sample(y, mu)
    ha, ra = run(a)
    hb, rb = run(b)
    with debug_switch("no_patch", 1):
        hc, rc = run(b)
    for u, v in ((ha, hb), (ra, rb), (ha, hc), (ra, rc)):
        assert float((u - v).abs().max()) <= 2e-5 * float(u.abs().max()), float((u - v).abs().max())
    assert not torch.equal(hb, hc), "no_patch changed nothing: the native encode did not run the first-layer patch kernel"


def _bench_line(*extra):
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"] and d["vs_baseline"] is None and d["n_gpus"] == 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] and r["ranking"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert "median" in r["ranking"]
    # the launch as the step issues it: reproducible from the committed profile files the object names
    i = r["in_step"]
    for k in ("launches_per_step", "meshes_per_launch", "algorithmic_bytes_per_launch", "avg_us", "achieved", "frac",
              "pmc_hbm_bytes_per_launch", "traffic_ratio", "kernel_stats", "pmc"):
        assert k in i, k
    assert i["launches_per_step"] * i["algorithmic_bytes_per_launch"] <= r["algorithmic_bytes_per_launch"]
    if i["avg_us"] is not None:          # a committed profile matches the kernel sources
        assert abs(i["frac"] - i["algorithmic_bytes_per_launch"] / (i["avg_us"] * 1e-6) / 1e9 / r["peak"]) < 1e-9
        assert os.path.exists(os.path.join(ROOT, i["kernel_stats"])) and os.path.exists(os.path.join(ROOT, i["pmc"]))
        if i["pmc_hbm_bytes_per_launch"]:
            assert abs(i["traffic_ratio"] - i["pmc_hbm_bytes_per_launch"] / i["algorithmic_bytes_per_launch"]) < 1e-9
            assert r["traffic"] == i["launches_per_step"] * i["pmc_hbm_bytes_per_launch"]
    else:
        assert i["frac"] is None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "meshes/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    return d


def test_bench_line_contract():
    """bench.py prints exactly one JSON line on stdout carrying the keys the driver reads, the roofline object of
    the dominant kernel and the CPU baseline (one child process: the script redirects its own stdout)."""
    d = _bench_line("--steps", "5", "--warmup", "2", "--prewarm-steps", "20")
    assert d["steps"] == 5 and d["warmup"] == 2 and d["prewarm_steps"] == 20 and d["unit"] == "meshes/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["config"]["workload"].startswith("configs[1]")
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    # 9 conv layers x (fwd, dX, dW) minus the first layer's dX and dW, minus one: the 5k level's 16 -> 16 stage takes both
    # gradients from ONE launch of the vertex-patch kernel (csrc/cheb_patch.hip)
    assert len(d["kernels"]) == 24 and any(k.startswith("conv dX+dW dec3") for k in d["kernels"])
    assert d["roofline"]["kernel"].startswith("k_patch_bwd<")
    # the other configurations, driver-timed by the same process after the headline's timed region
    v = d["variants"]
    assert set(v) == {"reference_loop", "bf16", "hires20k", "infer"}
    r = v["reference_loop"]                              # the reference's own loop (main.py:74-81,251), same protocol
    assert r["workload"].startswith("configs[1]") and "torch.optim.Adam" in r["optimizer"] and r["steps"] == 5
    assert abs(r["value"] - 64 * 1e3 / r["ms_per_step"]) < 1e-6 * r["value"] and 0 < r["step_roofline"]["frac"] < 1
    assert v["bf16"]["dtype"] == "bf16" and v["bf16"]["workload"].startswith("configs[1]") and v["bf16"]["steps"] == 5
    assert v["hires20k"]["workload"].startswith("configs[3]") and v["hires20k"]["value"] > 0
    for leg in (v["bf16"], v["hires20k"]):
        assert abs(leg["value"] - 64 * 1e3 / leg["ms_per_step"]) < 1e-6 * leg["value"] and 0 < leg["step_roofline"]["frac"] < 1
    # configs[4]: the three hipGraph replay latencies (ms), replay checked bitwise against eager inside the bench
    assert v["infer"]["workload"].startswith("configs[4]") and v["infer"]["replay_equals_eager_bitwise"] is True
    assert set(v["infer"]["latency_ms"]) == {"b1", "b32", "b256"} and all(0 < t < 50 for t in v["infer"]["latency_ms"].values())


def test_bench_infer_line_contract():
    """bench.py --config infer (BASELINE configs[4]): the same schema, latency-like (lower is better), with the three
    hipGraph replay latencies the configuration names."""
    d = _bench_line("--config", "infer", "--steps", "20", "--warmup", "3")
    assert d["unit"] == "ms" and d["higher_is_better"] is False and d["config"]["workload"].startswith("configs[4]")
    assert set(d["latency_ms"]) == {"b1", "b32", "b256"} and d["value"] == d["latency_ms"]["b32"]
    assert all(0.0 < v < 50.0 for v in d["latency_ms"].values())
    assert d["config"]["hipgraph"] is True and d["config"]["replay_equals_eager_bitwise"] is True


@pytest.mark.parametrize("B", [5, 16, 40, 64])
def test_round3_forms_against_their_debug_switches(B):
    """Every form this round added to the native step has a debug switch that restores the previous launch sequence; the
    step must not care: storing only what is read (keep_enc_out), the final layer's map inside the loss launch
    (no_final_fuse) and the stack kernel's shape (tstack_tall) change NOTHING, bit for bit; lazy rows between the final
    layer and the last decoder stage (no_src3) and the side the weight-gradient recurrence runs on (dw_tie_x) re-order
    fp32 sums: gradients within 2e-5 relative, forward outputs bitwise.  The level-0 lane (l0_lane: the 5k level's weight
    gradient held back behind the level's dX and cut into part-batch launches on the dense lane, B >= 16) moves launches
    only: bitwise against the single launch and against a three-way cut (B = 40: 16 + 16 + 8 meshes)."""
    from conftest import CFG_5K
    from meshvae_hip import debug_switch
    from meshvae_hip.engine import NativeStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, nn_[0], 3, generator=g).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, 16, generator=g).to(dev)

    def run(switch, slab_first_layer=False):
        torch.manual_seed(666)
        net = cheb_VAE(3, dict(CFG_5K), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
        key, val = switch if switch else ("keep_enc_out", 0)
        # (l0_lane_any: the level-0 lane at these batch sizes too -- by default it is taken for 56 < B <= 64 only)
        with debug_switch("l0_lane_any", 1), debug_switch("no_enc0_patch", 1 if slab_first_layer else 0), debug_switch(key, val):
            nat = NativeStep(net, B)
            drop_u = torch.rand(B * nat.u_cols, generator=torch.Generator().manual_seed(9)).to(dev)
            loss, corr, recon, (kld, rec, z_), yh = nat.forward_backward(x, x.double(), y, eps=eps, drop_u=drop_u)
            torch.cuda.synchronize()
        return (dict(loss=loss.clone(), recon=recon.clone(), kld=kld.clone(), rec=rec.clone(), z=z_.clone(), yh=yh.clone()),
                {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    base_out, base_g = run(None)
    # keep_enc_out and tstack_tall are forms of the SLAB first layer (cheb_lds.hip + k_cheb_tstack; since round 5 the step
    # runs k_patch_enc0 there, which has neither): they are held against that form (debug switch no_enc0_patch)
    slab_out, slab_g = run(None, slab_first_layer=True)
    for switch in (("keep_enc_out", 1), ("tstack_tall", 1)):
        out, grads = run(switch, slab_first_layer=True)
        assert all(torch.equal(out[k], slab_out[k]) for k in slab_out), switch
        assert all(torch.equal(grads[k], slab_g[k]) for k in slab_g), switch
    for switch, exact in ((("no_final_fuse", 1), True),
                          (("fork_small", 0), True),        # (every coarse layer's weight gradient behind a fork of its own)
                          (("l0_lane", 0), True),           # (the 5k level's weight gradient as ONE launch on the conv lane)
                          (("l0_lane", 3), True),           # (... in three part-batch launches: 24 + 24 + 16 meshes)
                          (("no_src3", 1), False), (("dw_tie_x", 1), False)):
        out, grads = run(switch)
        for k in base_out:
            assert torch.equal(out[k], base_out[k]), (switch, k)
        for k in base_g:
            if exact:
                assert torch.equal(grads[k], base_g[k]), (switch, k)
            else:
                den = float(base_g[k].norm())
                assert den == 0.0 or float((grads[k] - base_g[k]).norm()) / den < 2e-5, (switch, k)


def _ref_model(which, dev, dropout=0.0):
    from conftest import CFG_5K
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    cfg, topo = (TINY_CFG, "topology_tiny.npz") if which == "tiny" else (CFG_5K, "topology_5k.npz")
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", topo), dev)
    torch.manual_seed(666)
    return cheb_VAE(3, dict(cfg, dropout=dropout), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()


def test_module_path_sees_a_replaced_parameter_object():
    """the module path caches its parameter list and, per batch size, a native step with the gradient
    views.  A Parameter OBJECT replaced after the first call (net.cls.weight = nn.Parameter(...)) must not leave the fused
    step training the old tensor: the identity check in cheb_VAE._all_params_require_grad drops the caches, the next call
    reads the new weights and its gradient lands on the new object -- the same numbers as a fresh model holding them."""
    from meshvae_hip.engine import _Batch
    dev = torch.device("cuda:0")
    B, N = 4, 162
    x = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    for mode in ("autograd", "assign"):
        net = _ref_model("tiny", dev).train()
        net.dropout.p = 0.0
        net.grad_mode = mode
        eps = torch.randn(B, net.z, generator=torch.Generator().manual_seed(4)).to(dev)
        net._eps_provider = lambda B_, Z_, d_: eps
        net(_Batch(x), x, y, m_type="train")[0].backward()
        name, old = next((n, p) for n, p in net.named_parameters() if n.endswith("weight"))
        mod = net.get_submodule(name.rsplit(".", 1)[0])
        fresh = torch.nn.Parameter(old.detach() * 0.5 + 0.01)
        setattr(mod, name.rsplit(".", 1)[1], fresh)
        for p in net.parameters():
            p.grad = None
        loss = net(_Batch(x), x, y, m_type="train")[0]
        loss.backward()
        assert fresh.grad is not None and float(fresh.grad.abs().sum()) > 0, (mode, name)
        ref = _ref_model("tiny", dev).train()
        ref.dropout.p = 0.0
        ref._eps_provider = lambda B_, Z_, d_: eps
        ref.load_state_dict(net.state_dict())
        l2 = ref(_Batch(x), x, y, m_type="train")[0]
        l2.backward()
        assert float(loss.detach()) == float(l2.detach()), mode
        got = dict(net.named_parameters())
        for n, p in ref.named_parameters():
            if p.grad is not None:
                assert torch.equal(got[n].grad, p.grad), (mode, n)


@pytest.mark.parametrize("which,B", [("tiny", 6), ("5k", 4)])
def test_reference_loop_ends_where_trainstep_ends(which, B):
    """The reference's own loop over the drop-in modules (main.py:74-81,251: zero_grad -> net(data, x_gt, y) ->
    loss.backward() -> torch.optim.Adam.step()) and engine.TrainStep (flat buffers, fused Adam) are the same training:
    same host noise, three steps, parameters within 1e-6 -- with the asynchronous launcher (an opt-in, where the device has
    hipStreamWaitValue64) and without it, with the gradients going through autograd or assigned by the fused backward
    (net.grad_mode = "assign"); the four module-path runs agree bitwise (launcher and grad mode move host work only).
    The 1e-6 holds for all but a handful of the 5k model's 712 642 parameters: the gradients of the first step ARE bitwise
    equal (same kernels; tools/diag/ref_vs_trainstep.py), the two Adam implementations differ in the last bit of an update,
    and Adam's m / (sqrt(v) + eps) turns that into 1e-6 .. 1e-5 where a later gradient is ~1e-8 (measured: 6 entries of
    dec_lin_2.weight beyond 1e-6 after three steps, largest 5.8e-6) -- so the bar is 1e-6 for all but <= 1e-4 of a tensor's
    entries and 2e-5 for those."""
    import meshvae_hip
    from meshvae_hip.engine import TrainStep, _Batch
    dev = torch.device("cuda:0")
    N = 162 if which == "tiny" else 4998
    x = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    x_gt = x.double()
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    net = _ref_model(which, dev)
    step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=False)
    step.load(x, x, y)
    step.x_gt = x_gt
    torch.manual_seed(5)
    for _ in range(3):
        step.step()
    want = {k: v.detach().clone() for k, v in net.state_dict().items()}
    runs = {}
    for use_async, grad_mode in ((True, "autograd"), (False, "autograd"), (True, "assign"), (False, "assign")):
        net = _ref_model(which, dev)
        net.async_launch = use_async
        if grad_mode == "assign":          # the module's opt-in: .grad assigned by the fused backward
            net.grad_mode = "assign"
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
        torch.manual_seed(5)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            loss, correct, out, z, y_hat = net(_Batch(x), x_gt, y, m_type="train")
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        ent = next(iter(net._fused_cache.values()))
        assert (ent["launcher"] is not None) == (use_async and meshvae_hip.launcher(0) is not None)
        assert net.dec_lin_1.weight.grad is None and net.cheb[0].weight.grad is not None
        runs[use_async, grad_mode] = ({k: v.detach().clone() for k, v in net.state_dict().items()}, [float(l) for l in losses])
        for k, v in runs[use_async, grad_mode][0].items():
            dlt = (v - want[k]).abs()
            assert float(dlt.max()) <= 2e-5, (k, float(dlt.max()))
            assert int((dlt > 1e-6).sum()) <= max(0, int(1e-4 * dlt.numel())), (k, int((dlt > 1e-6).sum()))
    first = runs[True, "autograd"]
    for key, (sd, ls) in runs.items():      # launcher on / off, gradients through autograd / assigned: the same numbers, bit for bit
        assert ls == first[1], key
        assert all(torch.equal(sd[k], first[0][k]) for k in want), key
    if meshvae_hip.launcher(0) is not None:
        meshvae_hip.check(meshvae_hip.lib().mvh_launcher_sync(meshvae_hip.launcher(0)))


def test_module_outputs_are_fresh_and_gradients_survive_later_steps():
    """The module path hands out FRESH tensors (what the reference's autograd does): outputs and .grad of one step keep
    their values while later steps run, with or without zero_grad -- nothing is a window into a reused buffer."""
    from meshvae_hip.engine import _Batch
    dev = torch.device("cuda:0")
    B = 5
    net = _ref_model("tiny", dev, dropout=0.2)
    x = torch.randn(B, 162, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    torch.manual_seed(9)
    loss, correct, recon, (kld, rec, z_), y_hat = net(_Batch(x), x.double(), y, m_type="train")
    loss.backward()
    held = [t.detach() for t in (loss, correct, recon, kld, rec, z_, y_hat)]
    snap = [t.clone() for t in held]
    g_held = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    g_snap = {k: g.clone() for k, g in g_held.items()}
    for i in range(4):
        if i % 2:
            net.zero_grad(set_to_none=True)
        net(_Batch(x * (i + 2.0)), x.double(), y, m_type="train")[0].backward()
    torch.cuda.synchronize()
    for a, b in zip(held, snap):
        assert torch.equal(a, b)
    # the first two later steps ACCUMULATED into the held .grad tensors (no zero_grad); after set_to_none the parameters
    # got new .grad tensors and the held ones were left alone
    for k, g in g_held.items():
        assert not torch.equal(g, g_snap[k]), k
    later = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert all(later[k].data_ptr() != g_held[k].data_ptr() for k in later)


def test_async_launcher_reports_a_failed_job_and_leaves_nothing_blocked():
    """A bad call (workspace too small) is refused SYNCHRONOUSLY, on the caller's thread, like the plain entry point --
    nothing is queued.  A job that fails or THROWS on the worker thread still writes its ticket -- the caller's stream does
    not hang -- and the failure surfaces at the next launcher call with the worker's message."""
    import meshvae_hip
    from meshvae_hip.engine import NativeStep
    dev = torch.device("cuda:0")
    lch = meshvae_hip.launcher(0)
    if lch is None:
        pytest.skip("the asynchronous launcher is not in use on this device / in this environment")
    L = meshvae_hip.lib()
    meshvae_hip.check(L.mvh_launcher_sync(lch))
    net = _ref_model("tiny", dev)
    nat = NativeStep(net, 3, grads="external")
    x = torch.randn(3, 162, 3, device=dev)
    y = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]], device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    outs = (torch.empty((), **f32), torch.empty((), dtype=torch.int64, device=dev), torch.empty(3, 162, 3, **f32),
            torch.empty(3, **f32), torch.empty(3, **f32), torch.empty(3, 16, **f32), torch.empty(3, 2, **f32),
            torch.empty(3, 16, **f32), torch.empty(3, 16, **f32))
    good_bytes = nat.ws_bytes
    nat.ws_bytes = 1024
    with pytest.raises(meshvae_hip.MeshVaeHipError, match="workspace too small"):
        nat.run_forward(x, x, y, None, None, outs, lch)    # refused before anything is queued
    meshvae_hip.check(L.mvh_launcher_sync(lch))             # ... so nothing is pending and nothing was kept
    nat.ws_bytes = good_bytes
    st = torch.cuda.current_stream(dev).cuda_stream
    for mode, msg in ((1, "deliberate exception"), (2, "deliberate error code")):
        meshvae_hip.check(L.mvh_launcher_test_job(lch, st, mode))   # accepted: the failure happens on the worker
        marker = torch.ones(4, device=dev) * 3                      # later work on the caller's stream ...
        torch.cuda.synchronize()                                    # ... completes: the ticket was written
        assert float(marker.sum()) == 12.0
        with pytest.raises(meshvae_hip.MeshVaeHipError, match=msg):
            meshvae_hip.check(L.mvh_launcher_sync(lch))
    meshvae_hip.check(L.mvh_launcher_test_job(lch, st, 0))
    nat.run_forward(x, x, y, None, None, outs, lch)         # the launcher is usable again
    meshvae_hip.check(L.mvh_launcher_sync(lch))
    torch.cuda.synchronize()
    assert torch.isfinite(outs[2]).all()


def test_async_launcher_with_two_hardware_queues_and_many_streams():
    """the launcher's caller leaves a blocked value wait on its own stream; a weight-gradient lane that shared
    that stream's hardware queue would sit behind the wait while the job waits for the lane.  The lanes of a job therefore
    have the launcher's (highest) stream priority -- another queue pool.  Here the runtime gets TWO hardware queues and the
    application eight busy streams of its own, in a fresh child process; a hang fails the test through the timeout."""
    import subprocess
    import sys
    from conftest import PKG, ROOT
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, meshvae_hip
from test_gpu_engine import _ref_model
from meshvae_hip.engine import _Batch
dev = torch.device("cuda:0")
assert meshvae_hip.launcher(0) is not None, "launcher not in use"
streams = [torch.cuda.Stream(dev) for _ in range(8)]
junk = [torch.randn(1 << 20, device=dev) for _ in streams]
net = _ref_model("5k", dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
B = 8
x = torch.randn(B, 4998, 3, device=dev)
y = torch.nn.functional.one_hot(torch.arange(B) %% 2, 2).to(dev)
for it in range(12):
    for s_, j in zip(streams, junk):
        with torch.cuda.stream(s_):
            j.mul_(1.0001)
    opt.zero_grad()
    loss = net(_Batch(x), x.double(), y, m_type="train")[0]
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("OK", float(loss))
""" % (ROOT, PKG, ROOT)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2", GPU_STREAMOPS_CP_WAIT="1", MESHVAE_ASYNC="1")
    try:
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.fail("the module path under the asynchronous launcher hung with 2 hardware queues and 8 application streams")
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


def test_async_launcher_mixed_usage_equals_the_synchronous_path():
    """The module path under the asynchronous launcher in the ways a script mixes calls: train steps at changing batch sizes
    (more sizes than the entry cache holds: an evicted workspace must not be named by a queued job), no_grad evaluations in
    between, outputs read late, piecewise encoder / sample calls (synchronous, on the caller's stream, same workspace
    family).  Everything must equal the same sequence with the launcher off, bit for bit."""
    import meshvae_hip
    from meshvae_hip.engine import _Batch
    dev = torch.device("cuda:0")
    if meshvae_hip.launcher(0) is None:
        pytest.skip("the asynchronous launcher is not in use on this device / in this environment")
    g = torch.Generator().manual_seed(21)
    data = {B: (torch.randn(B, 162, 3, generator=g).to(dev), torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev))
            for B in (3, 5, 8, 2, 6)}

    def run(use_async):
        net = _ref_model("tiny", dev, dropout=0.2)
        net.async_launch = use_async
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
        torch.manual_seed(77)
        kept, out = [], []
        for it in range(14):
            B = (3, 5, 8, 2, 6)[it % 5]
            x, y = data[B]
            opt.zero_grad()
            loss, correct, recon, (kld, rec, z_), y_hat = net(_Batch(x), x.double(), y, m_type="train")
            kept.append((loss.detach(), recon, z_))                 # read only at the very end
            loss.backward()
            opt.step()
            if it % 4 == 3:
                net.eval()
                with torch.no_grad():
                    ev = net(_Batch(x), x, y, m_type="test")
                    h = net.encoder(x)
                    rec2 = net.sample(y, net.z_mean(torch.cat([y.float(), h], -1)))
                out.append((ev[0].clone(), ev[2].clone(), rec2.clone()))
                net.train()
        torch.cuda.synchronize()
        return ([tuple(t.clone() for t in k) for k in kept], out, {k: v.clone() for k, v in net.state_dict().items()})
    a, b = run(True), run(False)
    for ka, kb in zip(a[0], b[0]):
        assert all(torch.equal(u, v) for u, v in zip(ka, kb))
    for ea, eb in zip(a[1], b[1]):
        assert all(torch.equal(u, v) for u, v in zip(ea, eb))
    assert all(torch.equal(a[2][k], b[2][k]) for k in a[2])
    meshvae_hip.check(meshvae_hip.lib().mvh_launcher_sync(meshvae_hip.launcher(0)))
