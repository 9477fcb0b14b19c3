"""CPU, world_size 2, gloo: the data-parallel host logic of meshvae_hip.engine -- batch sharding,
the flat parameter/gradient buffers and the single sum all-reduce of the flat gradient.
(The compute itself is GPU-only; here each rank fills its flat gradient with a known function of
its shard, which is exactly what the backward pass does on the GPU.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, TINY_CFG


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from meshvae_hip.engine import FlatParams, rank_generators, shard_range
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), "cpu")
    # replicas must not depend on equal seeds: every rank initialises DIFFERENTLY here, rank 0's values win
    torch.manual_seed(666 + 17 * rank)
    net = cheb_VAE(3, TINY_CFG, D, U, A, nn_)
    flat = FlatParams(net)
    mine = flat.param.clone()
    assert flat.broadcast() is True
    torch.save(flat.param.clone(), os.path.join(out_dir, f"p{rank}.pt"))
    assert torch.equal(flat.param, mine) == (rank == 0)
    # per-rank noise streams (reparameterisation eps on the host generator): seed + rank
    host_gen, dev_gen = rank_generators(666, rank, "cpu")
    assert dev_gen is None
    torch.save(torch.normal(mean=0, std=1, size=(4, 16), generator=host_gen), os.path.join(out_dir, f"eps{rank}.pt"))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    # re-homing keeps values, names and order; params and grads are views of the flat buffers
    # (every tensor starts on a 256-byte boundary; the padding is zero in both buffers)
    assert flat.n_params == sum(v.numel() for v in before.values()) and flat.numel >= flat.n_params
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    for p, off in zip(net.parameters(), flat.offsets):
        assert off % 64 == 0
        assert p.data_ptr() == flat.param.data_ptr() + 4 * off and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * off
    assert float(flat.param.abs().sum()) == float(sum(v.abs().sum() for v in before.values()))
    # every rank owns a contiguous shard of the global batch
    G = 10
    lo, hi = shard_range(G, rank, world)
    sample_grads = torch.arange(G, dtype=torch.float32).view(G, 1) * torch.ones(1, flat.numel) + 0.5
    flat.zero_grad()
    for p, off in zip(net.parameters(), flat.offsets):   # "backward": accumulate into the .grad views
        p.grad += sample_grads[lo:hi].sum(0)[off:off + p.numel()].view_as(p)
    w = flat.all_reduce()
    assert w == world
    want = torch.zeros(flat.numel)
    for p, off in zip(net.parameters(), flat.offsets):
        want[off:off + p.numel()] = sample_grads.sum(0)[off:off + p.numel()]
    torch.testing.assert_close(flat.grad, want)     # sum over ALL samples == single-rank large batch
    torch.save(flat.grad.clone(), os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = (torch.load(os.path.join(str(tmp_path), f"g{r}.pt")) for r in (0, 1))
    assert torch.equal(g0, g1)                      # ranks agree bitwise after the all-reduce
    p0, p1 = (torch.load(os.path.join(str(tmp_path), f"p{r}.pt")) for r in (0, 1))
    assert torch.equal(p0, p1)                      # identical parameters after the broadcast from rank 0 ...
    torch.manual_seed(666)
    from meshvae_hip.engine import FlatParams
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), "cpu")
    assert torch.equal(FlatParams(cheb_VAE(3, TINY_CFG, D, U, A, nn_)).param, p0)   # ... namely rank 0's (seed 666)
    e0, e1 = (torch.load(os.path.join(str(tmp_path), f"eps{r}.pt")) for r in (0, 1))
    assert not torch.equal(e0, e1)                  # ranks draw different reparameterisation noise
    assert torch.equal(e0, torch.normal(mean=0, std=1, size=(4, 16), generator=torch.Generator().manual_seed(666)))


def test_no_grad_range_covers_dec_lin_1():
    """The flat-buffer span the fused Adam leaves untouched = dec_lin_1 (weight + bias), which the forward never
    uses (cheb_VAE.py:165) and torch.optim.Adam therefore never updates (its .grad stays None)."""
    from meshvae_hip.engine import FlatParams
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), "cpu")
    net = cheb_VAE(3, TINY_CFG, D, U, A, nn_)
    flat = FlatParams(net)
    lo, hi = flat.no_grad_range()
    iw, ib = flat.names.index("dec_lin_1.weight"), flat.names.index("dec_lin_1.bias")
    assert lo == flat.offsets[iw] and ib == iw + 1 and hi == flat.offsets[ib + 1]
    assert hi - lo >= net.dec_lin_1.weight.numel() + net.dec_lin_1.bias.numel()
    assert FlatParams(torch.nn.Linear(3, 4)).no_grad_range() == (0, 0)


def test_shard_range_partitions_the_batch():
    from meshvae_hip.engine import shard_range
    for G in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(G, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
