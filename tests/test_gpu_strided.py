"""GPU: strided [B, N, C] views across the module boundary (last row).

The reference's own modules hand transposed views to their arithmetic (nn/conv.py:560, nn/pool.py:18): a caller that
keeps activations vertex-major ([N, B, C]-physical) passes `x.transpose(0, 1)`.  `ChebConv_batch` / `SurfacePool` take
such a view WITHOUT the `.contiguous()` copy: the LDS-resident kernels read the rows in place through their row map
(mvh_cheb_conv_fwd_strided / _bwd_strided, mvh_pool_fwd_strided).  Held to the contiguous path on the same numbers:
forward and dX bitwise (same kernels, same order of operations); dW / db are asserted to 1e-6 relative (the
weight-gradient kernel takes its general row loop instead of the branch-free one) and printed: measured bitwise equal in
every case as well."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _lap(name, level, dev):
    from nn.conv import ChebConv_batch
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    n = int(z["num_nodes"][level])
    ei = torch.from_numpy(np.vstack([z[f"A{level}_row"], z[f"A{level}_col"]]).astype(np.int64)).to(dev)
    return ChebConv_batch.norm(ei, n) + (n,)


def _views(B, N, C, dev, seed):
    """the same numbers as (a) the transpose of an [N, B, C] tensor, (b) every other mesh of a 2B batch, (c) the leading C
    channels of a 2C-channel tensor -- and as one contiguous tensor"""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(B, N, C, generator=g).to(dev)
    a = base.transpose(0, 1).contiguous().transpose(0, 1)                 # [N, B, C]-physical
    b2 = torch.zeros(2 * B, N, C, device=dev)
    b2[::2] = base
    c2 = torch.zeros(B, N, 2 * C, device=dev)
    c2[..., :C] = base
    views = {"vertex-major": a, "batch slice": b2[::2], "channel slice": c2[..., :C]}
    for v in views.values():
        assert not v.is_contiguous() and torch.equal(v, base)
    return base, views


@pytest.mark.parametrize("topo,level,cin,cout,K,relu", [
    ("topology_tiny.npz", 0, 16, 16, 6, True),      # 162 vertices: one vertex per thread
    ("topology_5k.npz", 1, 16, 32, 6, True),        # 1250 vertices: two per thread, 32 output channels
    ("topology_5k.npz", 0, 16, 16, 6, True),        # the 4998-vertex level: 1024 x 5 forward / dX, 512 x 10 dW kernels
    ("topology_5k.npz", 0, 16, 16, 6, False),       # ... without the fused ReLU (mask-free backward)
    ("topology_5k.npz", 2, 32, 16, 3, True),        # 313 vertices, 32 input channels, K = 3
])
def test_cheb_conv_reads_strided_views_in_place(topo, level, cin, cout, K, relu):
    from nn.conv import ChebConv_batch
    dev = _dev()
    ei, nrm, N = _lap(topo, level, dev)
    B = 5
    base, views = _views(B, N, cin, dev, seed=level + cin)
    g = torch.Generator().manual_seed(99)
    w, b = (torch.randn(K, cin, cout, generator=g) * 0.1).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
    gy = torch.randn(B, N, cout, generator=g).to(dev)

    def run(x):
        conv = ChebConv_batch(cin, cout, K).to(dev)
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.bias.copy_(b)
        x = x.detach().requires_grad_(True)          # (a leaf with the view's strides)
        assert x.stride() == x.detach().stride()
        y = conv(x, ei, nrm, relu=relu)
        took_view = y.grad_fn.view is not None
        y.backward(gy)
        return y.detach(), x.grad, conv.weight.grad, conv.bias.grad, took_view
    y0, dx0, dw0, db0, v0 = run(base)
    assert not v0
    for name, xv in views.items():
        xs = torch.empty_strided(xv.shape, xv.stride(), device=dev)      # a fresh tensor with the view's layout
        xs.copy_(xv)
        y1, dx1, dw1, db1, v1 = run(xs)
        assert v1, f"{name}: the strided entry was not taken (copy fallback)"
        assert torch.equal(y1, y0), name
        assert torch.equal(dx1, dx0), name
        torch.testing.assert_close(dw1, dw0, rtol=1e-6, atol=1e-6 * float(dw0.abs().max()))
        torch.testing.assert_close(db1, db0, rtol=1e-6, atol=1e-6 * float(db0.abs().max()))
        print(f"[{topo} level {level} {cin}->{cout} K={K}] {name}: dW bitwise {torch.equal(dw1, dw0)}, db bitwise {torch.equal(db1, db0)}")


def test_unsupported_shapes_fall_back_to_the_copy():
    """A level the LDS-resident kernels do not take (20 164 vertices): the library answers MVH_ERR_UNSUPPORTED and the
    wrapper copies; a view whose rows are not 16-byte aligned (channels 1..3 of a 6-channel tensor) does not qualify in
    the first place.  Either way the result is the contiguous path's."""
    from nn.conv import ChebConv_batch
    from test_gpu_parity import _grid_mesh_edges
    dev = _dev()
    side = 142
    N = side * side
    ei, nrm = ChebConv_batch.norm(torch.from_numpy(_grid_mesh_edges(side)).to(dev), N)
    conv = ChebConv_batch(16, 16, 4).to(dev)
    x = torch.randn(N, 2, 16, device=dev).transpose(0, 1)
    y = conv(x, ei, nrm)
    assert y.grad_fn.view is None                       # the big level refused, the copy ran
    torch.testing.assert_close(y, conv(x.contiguous(), ei, nrm), rtol=0, atol=0)
    ei5, nrm5, N5 = _lap("topology_5k.npz", 0, dev)
    conv3 = ChebConv_batch(3, 16, 6).to(dev)
    x6 = torch.randn(2, N5, 6, device=dev)
    y3 = conv3(x6[..., 1:4], ei5, nrm5)
    assert y3.grad_fn.view is None
    assert torch.equal(y3, conv3(x6[..., 1:4].contiguous(), ei5, nrm5))
    y3b = conv3(x6[..., :3], ei5, nrm5)                 # (the aligned slice IS read in place: 12-byte rows at a 24-byte stride)
    assert y3b.grad_fn.view is not None
    assert torch.equal(y3b, conv3(x6[..., :3].contiguous(), ei5, nrm5))


def test_first_layer_three_channels_vertex_major():
    """3 input channels (cheb.0): rows of 12 bytes; the vertex-major view qualifies when its base is 16-byte aligned."""
    from nn.conv import ChebConv_batch
    dev = _dev()
    ei, nrm, N = _lap("topology_5k.npz", 0, dev)
    conv = ChebConv_batch(3, 16, 6).to(dev)
    xp = torch.randn(N, 4, 3, device=dev)
    x = xp.transpose(0, 1)
    xa = x.detach().requires_grad_(True)
    ya = conv(xa, ei, nrm, relu=True)
    xb = x.contiguous().requires_grad_(True)
    yb = conv(xb, ei, nrm, relu=True)
    assert ya.grad_fn.view is not None
    assert torch.equal(ya, yb)
    gy = torch.randn_like(ya)
    ga = torch.autograd.grad(ya, [xa] + list(conv.parameters()), gy)
    gb = torch.autograd.grad(yb, [xb] + list(conv.parameters()), gy)
    assert torch.equal(ga[0], gb[0])
    for a, b in zip(ga[1:], gb[1:]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6 * float(b.abs().max()))


@pytest.mark.parametrize("which", ["D0", "U0", "U1"])
def test_surface_pool_reads_strided_views_in_place(which, topotiny_npz):
    from nn.pool import SurfacePool
    dev = _dev()
    z = topotiny_npz
    idx = torch.from_numpy(np.vstack([z[f"{which}_row"], z[f"{which}_col"]]).astype(np.int64)).to(dev)
    shape = tuple(int(v) for v in z[f"{which}_shape"])
    P = torch.sparse_coo_tensor(idx, torch.from_numpy(z[f"{which}_val"]).to(dev), shape, check_invariants=False)
    pool = SurfacePool()
    for C in (3, 16):
        base, views = _views(4, shape[1], C, dev, seed=C)
        y0 = pool(base, P)
        for name, xv in views.items():
            xs = torch.empty_strided(xv.shape, xv.stride(), device=dev)
            xs.copy_(xv)
            xs.requires_grad_(True)
            y1 = pool(xs, P)
            assert torch.equal(y1, y0), (which, C, name)
            y1.backward(torch.ones_like(y1))
            xb = base.clone().requires_grad_(True)
            pool(xb, P).backward(torch.ones_like(y0))
            assert torch.equal(xs.grad, xb.grad)


def test_strided_entry_argument_checks():
    """Strides that are not multiples of Cin, a zero mesh stride, a short workspace: MVH_ERR_INVALID with a message, before
    any launch."""
    import ctypes
    from meshvae_hip import lib, topology
    dev = _dev()
    ei, nrm, N = _lap("topology_tiny.npz", 0, dev)
    op = topology.laplacian(ei, nrm, N)
    L = lib()
    x = torch.randn(N, 2, 16, device=dev)
    w, out = torch.randn(6, 16, 16, device=dev), torch.empty(2, N, 16, device=dev)
    nb = L.mvh_cheb_conv_strided_ws_bytes(2, N, 16, 16, 6)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    args = lambda ms, rs, n_ws: (st, op.fwd.ref, x.data_ptr(), ms, rs, w.data_ptr(), None, out.data_ptr(), None, 2, N, 16, 16, 6, 0,
                                 ws.data_ptr(), n_ws)
    assert L.mvh_cheb_conv_fwd_strided(*args(16, 24, nb)) == 1 and b"multiples of Cin" in L.mvh_last_error()
    assert L.mvh_cheb_conv_fwd_strided(*args(0, 32, nb)) == 1 and b"mesh stride 0" in L.mvh_last_error()
    assert L.mvh_cheb_conv_fwd_strided(*args(16, 32, 128)) == 1 and b"workspace too small" in L.mvh_last_error()
    assert L.mvh_cheb_conv_fwd_strided(*args(16, 32, nb)) == 0
    torch.cuda.synchronize()
    from nn.conv import ChebConv_batch
    conv = ChebConv_batch(16, 16, 6).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.zero_()
    assert torch.equal(out, conv(x.transpose(0, 1).contiguous(), ei, nrm))
