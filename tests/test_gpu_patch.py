"""GPU: the vertex-patch ChebConv kernels (csrc/cheb_patch.hip, plans from meshvae_hip/patches.py) against the CPU oracle
and against the slab kernels they replace.

A level of 2 049 .. 5 119 vertices whose graph cuts into patches that fit a CU (the 5k hip-bone template: two bones, four
patches; a 5k torus: five) runs its 16 -> 16 layers as (mesh, vertex patch) workgroups with the K Cin x Cout contraction
on v_mfma_f32_16x16x4_f32.  Bars: forward within 1e-4 absolute of the oracle (nn/conv.py:557-577's contract), gradients
within 1e-4 relative; the plan with fused pooling rows and the plain plan agree to rounding (a vertex's neighbours are summed
in a per-plan order: the lists are permuted against LDS bank conflicts)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _edges(npz):
    ei = torch.from_numpy(np.stack([npz["A0_row"], npz["A0_col"]]).astype(np.int64))
    return ei, int(npz["num_nodes"][0])


def _conv_case(ei_cpu, N, K, relu, B=3, seed=0, no_patch=False):
    import meshvae_hip
    from nn.conv import ChebConv_batch
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, 16, generator=g)
    w = torch.randn(K, 16, 16, generator=g) * 0.1
    b = torch.randn(16, generator=g) * 0.1
    gy = torch.randn(B, N, 16, generator=g)
    ei, nrm = ChebConv_batch.norm(ei_cpu.to(dev), N)
    conv = ChebConv_batch(16, 16, K).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(b)
    xd = x.to(dev).requires_grad_(True)
    with meshvae_hip.debug_switch("no_patch", 1 if no_patch else 0):
        y = conv(xd, ei, nrm, relu=relu)
        y.backward(gy.to(dev))
    torch.cuda.synchronize()
    return (x, w, b, gy), (y.detach().cpu(), xd.grad.cpu(), conv.weight.grad.cpu(), conv.bias.grad.cpu())


def _oracle(ei_cpu, N, x, w, b, gy, relu):
    from oracle import cheb_oracle as O
    eio, nrmo = O.cheb_norm(ei_cpu, N)
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    yo = O.cheb_conv(xo, eio, nrmo, wo, bo)
    if relu:
        yo = torch.relu(yo)
    yo.backward(gy)
    return yo.detach(), xo.grad, wo.grad, bo.grad


def _has_plan(ei_cpu, N):
    from meshvae_hip import topology
    from nn.conv import ChebConv_batch
    dev = torch.device("cuda:0")
    ei, nrm = ChebConv_batch.norm(ei_cpu.to(dev), N)
    op = topology.laplacian(ei, nrm, N)
    return bool(op.fwd.struct.patch), op


@pytest.mark.parametrize("fixture,K,relu,B", [("topology_5k.npz", 6, True, 3), ("topology_5k.npz", 6, False, 9),
                                              ("topology_5k.npz", 3, True, 2), ("topology_5k.npz", 1, True, 2),
                                              ("hier_torus5k.npz", 6, True, 3), ("hier_torus5k.npz", 5, False, 1)])
def test_patch_conv_matches_oracle_and_slab_kernels(fixture, K, relu, B):
    from conftest import load_golden
    ei_cpu, N = _edges(load_golden(fixture))
    has, _ = _has_plan(ei_cpu, N)
    assert has, "the level must carry a patch plan (meshvae_hip/patches.py)"
    ins, got = _conv_case(ei_cpu, N, K, relu, B=B, seed=K + B)
    ref = _oracle(ei_cpu, N, *ins, relu)
    _, slab = _conv_case(ei_cpu, N, K, relu, B=B, seed=K + B, no_patch=True)
    y, dx, dw, db = got
    torch.testing.assert_close(y, ref[0], rtol=0, atol=1e-4)
    if relu:
        assert torch.equal(y > 0, ref[0] > 0)
    torch.testing.assert_close(dx, ref[1], rtol=1e-4, atol=1e-4)
    sw, sb = float(ref[2].abs().max()), float(ref[3].abs().max())
    torch.testing.assert_close(dw, ref[2], rtol=1e-4, atol=1e-5 * sw + 1e-5)
    torch.testing.assert_close(db, ref[3], rtol=1e-4, atol=1e-5 * sb + 1e-5)
    # ... and the kernels it replaces (fp32 reassociation only)
    torch.testing.assert_close(y, slab[0], rtol=0, atol=2e-5)
    torch.testing.assert_close(dx, slab[1], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dw, slab[2], rtol=1e-4, atol=1e-5 * sw + 1e-5)


def test_patch_kernels_really_ran():
    """The switch `no_patch` changes which kernels run: with the same inputs the two paths agree to rounding but not
    bitwise (different summation trees) -- a silent fallback to the slab kernels would make them identical."""
    from conftest import load_golden
    ei_cpu, N = _edges(load_golden("topology_5k.npz"))
    _, a = _conv_case(ei_cpu, N, 6, True, B=2, seed=5)
    _, b = _conv_case(ei_cpu, N, 6, True, B=2, seed=5, no_patch=True)
    assert not torch.equal(a[0], b[0])
    torch.testing.assert_close(a[0], b[0], rtol=0, atol=2e-5)


def test_plan_with_pooling_rows_matches_spmm_of_plain_dx():
    """The step engine's form: dX pooled by U^T inside the kernel (plan built with the level's un-pooling operator)
    against pool_bwd of the plain kernel's dX, and both plans' dW."""
    import meshvae_hip
    from conftest import load_golden
    from meshvae_hip import lib, check, topology
    from nn.conv import ChebConv_batch
    npz = load_golden("topology_5k.npz")
    ei_cpu, N = _edges(npz)
    dev = torch.device("cuda:0")
    ei, nrm = ChebConv_batch.norm(ei_cpu.to(dev), N)
    lap = topology.laplacian(ei, nrm, N)
    U = torch.sparse_coo_tensor(torch.from_numpy(np.stack([npz["U0_row"], npz["U0_col"]]).astype(np.int64)),
                                torch.from_numpy(npz["U0_val"]), tuple(int(v) for v in npz["U0_shape"])).to(dev)
    up = topology.pool_operator(U)
    got = topology.patch_plan(lap, 5, up)
    assert got is not None and got[0].n_pool_rows == up.n_in
    B, K = 5, 6
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, 16, generator=g).to(dev)
    w = (torch.randn(K, 16, 16, generator=g) * 0.1).to(dev)
    gy = torch.randn(B, N, 16, generator=g).to(dev)
    signs = (torch.rand(B, N, 4, generator=g) * 16).to(torch.uint8).to(dev)
    L = lib()
    ws_bytes = L.mvh_cheb_conv_bwd_ws_bytes(B, N, 16, 16, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(patch_ptr):
        fwd, bwd = meshvae_hip.CsrStruct.from_buffer_copy(lap.fwd.struct), meshvae_hip.CsrStruct.from_buffer_copy(lap.bwd.struct)
        fwd.patch = bwd.patch = patch_ptr
        dx, dw, db = torch.empty(B, N, 16, device=dev), torch.empty(K, 16, 16, device=dev), torch.empty(16, device=dev)
        check(L.mvh_cheb_conv_bwd_signs(st, ctypes.byref(fwd), ctypes.byref(bwd), x.data_ptr(), w.data_ptr(), None,
                                        signs.data_ptr(), gy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                        B, N, 16, 16, K, ws.data_ptr(), ws_bytes))
        torch.cuda.synchronize()
        return dx, dw, db
    dx_a, dw_a, db_a = run(lap.fwd.struct.patch)                  # plain plan
    dx_b, dw_b, db_b = run(ctypes.addressof(got[0]))              # plan with pooling rows (dx not pooled by this entry)
    torch.testing.assert_close(dx_a, dx_b, rtol=1e-5, atol=1e-5)     # (per-plan order of a vertex's neighbour sums)
    torch.testing.assert_close(dw_a, dw_b, rtol=1e-4, atol=1e-5 * float(dw_a.abs().max()) + 1e-5)
    torch.testing.assert_close(db_a, db_b, rtol=1e-5, atol=1e-6)
    with meshvae_hip.debug_switch("no_patch", 1):
        dx_c, dw_c, db_c = run(None)
    torch.testing.assert_close(dx_a, dx_c, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dw_a, dw_c, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db_a, db_c, rtol=1e-4, atol=1e-4)


def _step_5k(switches, B=5, storage="f32", seed=21):
    """one NativeStep forward + backward of the 5k model under debug switches -> (loss, recon, z, gradients)"""
    import contextlib
    import os
    import sys
    import meshvae_hip
    from conftest import CFG_5K
    from meshvae_hip.engine import NativeStep
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    dev = torch.device("cuda:0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    D, U, A, nn_ = load_topology(os.path.join(root, "tests", "golden", "topology_5k.npz"), dev)
    torch.manual_seed(666)
    net = cheb_VAE(3, dict(CFG_5K, dropout=0.0), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4998, 3, generator=g).to(dev)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    eps = torch.randn(B, net.z, generator=g).to(dev)
    with contextlib.ExitStack() as es:
        for k, v in switches.items():
            es.enter_context(meshvae_hip.debug_switch(k, v))
        nat = NativeStep(net, B, storage=storage)
        outs = []
        for _ in range(2):          # (twice: the second forward reuses the workspace whose stack the first backward consumed)
            loss, _, recon, (_, _, z_), _ = nat.forward_backward(x, x, y, eps=eps, drop_u=None)
            torch.cuda.synchronize()
            outs.append((float(loss), recon.clone(), z_.clone(),
                         {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    assert len(outs[0][3]) > 10 and all(torch.equal(outs[0][3][k], outs[1][3][k]) for k in outs[0][3])
    return outs[1]


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_first_layer_patch_kernel_in_the_step(storage):
    """k_patch_enc0 (the 3 -> 16 layer's recurrence on the input side, pooled rows + sign bytes + the weight gradient's
    T_k stack out of one launch) against the slab kernel + k_cheb_tstack it replaces in the step (debug switch
    no_enc0_patch): the same step to fp32 reassociation -- and not bitwise, so the kernel really ran.  The first layer's
    weight gradient is the product of exactly the two things the kernel hands over (stack and sign bytes)."""
    a = _step_5k({}, storage=storage)
    b = _step_5k({"no_enc0_patch": 1}, storage=storage)
    tol = 2e-5 if storage == "f32" else 2e-2      # (bf16 storage: a stored value may land on the other side of a rounding boundary)
    assert abs(a[0] - b[0]) <= (1e-6 if storage == "f32" else 1e-3) * abs(b[0])
    scale = float(b[1].abs().max())
    assert float((a[1] - b[1]).abs().max()) <= tol * scale
    assert not all(torch.equal(a[3][k], b[3][k]) for k in a[3]), "the switch changed nothing: the patch kernel did not run"
    for k in b[3]:
        rel = float((a[3][k] - b[3][k]).norm() / b[3][k].norm().clamp_min(1e-20))
        assert rel < (1e-4 if storage == "f32" else 5e-2), (k, rel)


def test_unpooling_inside_the_patch_forward_is_bitwise_the_stored_form():
    """The last decoder stage's forward takes the COARSE tensor and un-pools it while it loads (plan rows `urec`, three taps
    per vertex in the pooling op's arithmetic); the stage before it then stores no un-pooled rows.  Against the stored form
    (debug switch no_patch_unpool: the previous stage's kernel writes U x from its epilogue) the whole step is the same
    bit for bit -- outputs and every gradient -- and the un-pooled tensor the backward reads is the one the forward wrote."""
    a = _step_5k({})
    b = _step_5k({"no_patch_unpool": 1})
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in b[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def test_final_layer_map_out_of_the_patch_forward_is_bitwise_the_loss_launch_form():
    """The final conv off its 20-vertex block is the per-vertex map x16 W_eff (cheb_VAE.py:288).  The last decoder stage's
    patch kernel writes it from its epilogue (the vertex's four quads collected in one lane, k_cheb_contract's 16-term fma
    chain), so neither a map launch nor the loss launch reads that stage's 20 MB output again.  Against the form that
    computes it inside the loss launch (debug switch no_patch_map) the step is the same bit for bit."""
    a = _step_5k({})
    b = _step_5k({"no_patch_map": 1})
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in b[3]:
        assert torch.equal(a[3][k], b[3][k]), k
