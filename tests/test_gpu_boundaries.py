"""GPU: ChebConv_batch at the sizes where the library changes kernels, on seeded random graphs, against the CPU oracle.

The LDS-resident kernels pick their shape from N + 1: <= 1024 one vertex per thread, <= 2048 two per thread, <= 5120 the
160 KB level-0 shapes (1024 x 5 forward / dX, 512 x 10 dW; 16 input channels at most), above that the streaming
pipeline (cheb_big.hip + contraction; rows longer than 8 neighbours: the K - 1 SpMM launches).  A row of more than 8
neighbours takes the ELL-overflow path on the small shapes (up to 12) and the general pipeline beyond.  Every case runs
the forward and all three gradients (fused ReLU included) against oracle/cheb_oracle.py at the 1e-4 bars of the other
parity tests; the graphs are rings with random chords, so vertex ids far apart are neighbours (no locality to hide an
indexing slip), one vertex is isolated, and B = 3 is not a multiple of the 8 XCDs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _random_graph(N, max_deg, seed, isolated=True):
    """Symmetric edge list [2, E]: a ring over the first N - 1 vertices plus random chords, every degree <= max_deg;
    vertex N - 1 stays isolated when asked."""
    g = np.random.default_rng(seed)
    n = N - 1 if isolated else N
    deg = np.full(N, 0)
    edges = set()
    for i in range(n):
        j = (i + 1) % n
        if i != j and (min(i, j), max(i, j)) not in edges:
            edges.add((min(i, j), max(i, j)))
            deg[i] += 1
            deg[j] += 1
    tries = 3 * n
    a, b = g.integers(0, n, tries), g.integers(0, n, tries)
    for i, j in zip(a.tolist(), b.tolist()):
        if i == j or deg[i] >= max_deg or deg[j] >= max_deg or (min(i, j), max(i, j)) in edges:
            continue
        edges.add((min(i, j), max(i, j)))
        deg[i] += 1
        deg[j] += 1
    # a few vertices pushed to exactly max_deg (the longest rows are the interesting ones)
    hubs = g.choice(n, size=min(8, n), replace=False)
    for h in hubs.tolist():
        for j in g.permutation(n).tolist():
            if deg[h] >= max_deg:
                break
            if j == h or deg[j] >= max_deg or (min(h, j), max(h, j)) in edges:
                continue
            edges.add((min(h, j), max(h, j)))
            deg[h] += 1
            deg[j] += 1
    e = np.array(sorted(edges), dtype=np.int64)
    order = g.permutation(2 * len(e))                      # edge order is the accumulation order: shuffle it
    both = np.concatenate([e, e[:, ::-1]])[order]
    assert deg.max() <= max_deg and (not isolated or deg[N - 1] == 0)
    return torch.from_numpy(np.ascontiguousarray(both.T)), int(deg.max())


CASES = [
    # N,   max_deg, Cin, Cout, K, relu
    (1023, 8, 16, 16, 6, True),      # N + 1 = 1024: the last one-vertex-per-thread size
    (1024, 8, 16, 32, 6, True),      # first two-per-thread size, 32 output channels
    (2047, 8, 32, 16, 6, True),      # N + 1 = 2048: the last two-per-thread size, 32 input channels
    (2048, 8, 16, 16, 6, True),      # first level-0 shape (1024 x 5 / 512 x 10)
    (2048, 8, 32, 16, 3, False),     # ... which does not exist for 32 input channels: general pipeline
    (5119, 8, 16, 16, 6, True),      # N + 1 = 5120: the largest mesh of the level-0 kernels (every slot a vertex)
    (5119, 8, 3, 16, 6, True),       # ... first-layer form (3 channels)
    (5120, 8, 16, 16, 6, True),      # one vertex more: the streaming kernels (pair-major stack, two-gradient pass)
    (5120, 8, 3, 16, 4, False),      # ... 3 channels, no ReLU
    (700, 12, 16, 16, 6, True),      # rows of up to 12 neighbours: ELL overflow words, one vertex per thread
    (1500, 11, 16, 32, 6, True),     # ... two vertices per thread
    (900, 15, 16, 16, 6, True),      # rows too long for the overflow words: general pipeline
    (3000, 10, 16, 16, 6, True),     # level-0 size with rows > 8: the level-0 shapes refuse, general pipeline
]


@pytest.mark.parametrize("N,max_deg,cin,cout,K,relu", CASES)
def test_cheb_conv_at_kernel_boundaries_matches_oracle(N, max_deg, cin, cout, K, relu):
    from nn.conv import ChebConv_batch
    from oracle import cheb_oracle as O
    dev = torch.device("cuda:0")
    ei_cpu, dmax = _random_graph(N, max_deg, seed=N + max_deg)
    assert dmax == max_deg
    B = 3
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, N, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) * 0.1
    b = torch.randn(cout, generator=g) * 0.1
    gy = torch.randn(B, N, cout, generator=g)
    eio, nrmo = O.cheb_norm(ei_cpu, N)
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    yo = O.cheb_conv(xo, eio, nrmo, wo, bo)
    if relu:
        yo = torch.relu(yo)
    yo.backward(gy)
    ei, nrm = ChebConv_batch.norm(ei_cpu.to(dev), N)
    conv = ChebConv_batch(cin, cout, K).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(b)
    xd = x.to(dev).requires_grad_(True)
    y = conv(xd, ei, nrm, relu=relu)
    y.backward(gy.to(dev))
    torch.testing.assert_close(y.detach().cpu(), yo.detach(), rtol=0, atol=1e-4)
    # ReLU ties: an element whose pre-activation is within fp32 noise of zero may flip between the two
    # implementations; its gradient contribution is then O(1) -- none occur at these seeds (asserted)
    if relu:
        assert torch.equal(y.detach().cpu() > 0, yo.detach() > 0)
    torch.testing.assert_close(xd.grad.cpu(), xo.grad, rtol=1e-4, atol=1e-4)
    scale_w, scale_b = float(wo.grad.abs().max()), float(bo.grad.abs().max())
    torch.testing.assert_close(conv.weight.grad.cpu(), wo.grad, rtol=1e-4, atol=1e-5 * scale_w + 1e-5)
    torch.testing.assert_close(conv.bias.grad.cpu(), bo.grad, rtol=1e-4, atol=1e-5 * scale_b + 1e-5)
    # the isolated vertex: L x = 0 there, so out = act(x (W_0 - W_2 + W_4 - ...) + b) exactly as the quirk rows
    we = sum(((-1.0) ** (k // 2)) * w[k] for k in range(0, K, 2))
    want = x[:, N - 1] @ we + b
    if relu:
        want = torch.relu(want)
    torch.testing.assert_close(y.detach().cpu()[:, N - 1], want, rtol=0, atol=1e-5)


def test_empty_batch_and_single_mesh():
    """B = 0 (an empty batch: nothing launched, empty output, zero weight gradients) and B = 1."""
    from nn.conv import ChebConv_batch
    from oracle import cheb_oracle as O
    dev = torch.device("cuda:0")
    N = 300
    ei_cpu, _ = _random_graph(N, 8, seed=1)
    ei, nrm = ChebConv_batch.norm(ei_cpu.to(dev), N)
    conv = ChebConv_batch(16, 16, 6).to(dev)
    x0 = torch.zeros(0, N, 16, device=dev, requires_grad=True)
    y0 = conv(x0, ei, nrm, relu=True)
    assert y0.shape == (0, N, 16)
    y0.sum().backward()
    assert x0.grad.shape == (0, N, 16)
    assert float(conv.weight.grad.abs().sum()) == 0 and float(conv.bias.grad.abs().sum()) == 0
    from nn.pool import SurfacePool
    P = torch.sparse_coo_tensor(torch.tensor([[0, 1, 2], [5, 7, 9]], device=dev), torch.ones(3, device=dev), (3, N))
    p0 = SurfacePool()(torch.zeros(0, N, 16, device=dev, requires_grad=True), P)
    assert p0.shape == (0, 3, 16)
    p0.sum().backward()
    x1 = torch.randn(1, N, 16)
    y1 = conv(x1.to(dev), ei, nrm)
    eio, nrmo = O.cheb_norm(ei_cpu, N)
    yo = O.cheb_conv(x1, eio, nrmo, conv.weight.detach().cpu(), conv.bias.detach().cpu())
    torch.testing.assert_close(y1.detach().cpu(), yo, rtol=0, atol=1e-4)
