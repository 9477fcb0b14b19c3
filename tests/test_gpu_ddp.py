"""GPU (-m gpu): the REAL data-parallel engine path with world_size 2 on the one GPU of the box.

Two fresh child processes share cuda:0 and form a gloo process group over CUDA tensors; each builds the model from a
DIFFERENT seed and drives `engine.TrainStep(group=...)` on its shard of a global batch for three optimizer steps.  That
executes, in the product code and with world > 1: FlatParams.broadcast (rank 0's parameters win), the per-step flat
gradient all-reduce between backward and optimizer, grad_scale = 1/world inside the fused Adam kernel, and the per-rank
noise generators (seed + rank).  The parent then runs ONE process on the whole batch with the same noise and compares.
RCCL itself needs one GPU per rank and is the driver's to run (SCALE_rNN.json); everything above it is covered here.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, TINY_CFG

pytestmark = pytest.mark.gpu
G, WORLD, STEPS, SEED = 8, 2, 3, 666


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(G, 162, 3, generator=g)
    y = torch.nn.functional.one_hot(torch.arange(G) % 2, 2)
    return x, y


def _net(dev, seed, dropout):
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_tiny.npz"), dev)
    torch.manual_seed(seed)
    return cheb_VAE(3, dict(TINY_CFG, dropout=dropout), D, U, A, nn_).to(dev).train()


def _worker(rank, world, port, out_dir, dropout, overlap):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from meshvae_hip.engine import TrainStep, shard_range
    net = _net(dev, SEED + 17 * rank, dropout)              # replicas must NOT rely on equal seeds
    lo, hi = shard_range(G, rank, world)
    step = TrainStep(net, hi - lo, lr=1e-3, weight_decay=5e-4, use_graph=False, group=dist.group.WORLD, noise_seed=SEED,
                     overlap_allreduce=overlap)
    assert step.world == world
    x, y = _data()
    step.load(x[lo:hi].to(dev), x[lo:hi].to(dev), y[lo:hi].to(dev))
    losses = []
    for _ in range(STEPS):
        loss, correct, recon = step.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    torch.save({"param": step.flat.param.cpu(), "grad": step.flat.grad.cpu(), "losses": losses,
                "adam_steps": int(step.opt.step_count)}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_trainstep_world2_on_one_gpu_equals_the_single_process_step(tmp_path, overlap):
    """dropout 0 (the reparameterisation noise is the only randomness, and the parent can rebuild it: rank r draws
    from Generator(seed + r)): three steps on two shards of 4 == three steps of one process on all 8 meshes."""
    port = _free_port()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), 0.0, overlap), nprocs=WORLD, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(WORLD))
    assert torch.equal(r0["param"], r1["param"])           # replicas agree bitwise after three all-reduced steps
    assert torch.equal(r0["grad"], r1["grad"])             # ... on the reduced gradient too
    assert r0["adam_steps"] == r1["adam_steps"] == STEPS
    assert r0["losses"] != r1["losses"]                    # (different shards, different noise)
    from meshvae_hip.engine import TrainStep
    dev = torch.device("cuda:0")
    net = _net(dev, SEED, 0.0)                             # rank 0's initial parameters (its seed)
    big = TrainStep(net, G, lr=1e-3, weight_decay=5e-4, use_graph=False)
    x, y = _data()
    big.load(x.to(dev), x.to(dev), y.to(dev))
    gens = [torch.Generator().manual_seed(SEED + r) for r in range(WORLD)]
    Z = net.z

    def eps_of_all_ranks():                                # what the two ranks drew for their shards, in shard order
        big.eps.copy_(torch.cat([torch.normal(mean=0, std=1, size=(G // WORLD, Z), generator=g) for g in gens]))
    big._draw_eps = eps_of_all_ranks
    losses = [float(big.step()[0]) for _ in range(STEPS)]
    torch.cuda.synchronize()
    # the single-process loss is the mean over 8 meshes = the mean of the two ranks' means over 4
    for k in range(STEPS):
        assert abs(losses[k] - 0.5 * (r0["losses"][k] + r1["losses"][k])) <= 2e-6 * abs(losses[k]) + 1e-3
    torch.testing.assert_close(r0["param"], big.flat.param.cpu(), rtol=1e-4, atol=2e-6)
    assert losses[-1] < losses[0]


def test_trainstep_world2_with_dropout_keeps_replicas_identical(tmp_path):
    """dropout 0.2: every rank draws its own masks (device generator seeded seed + rank); the replicas must still hold
    bitwise identical parameters after every all-reduced step, and the loss must fall."""
    port = _free_port()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), 0.2, False), nprocs=WORLD, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(WORLD))
    assert torch.equal(r0["param"], r1["param"]) and torch.isfinite(r0["param"]).all()
    assert all(l == l for l in r0["losses"] + r1["losses"])


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
    environment), rehearsed on the one GPU of this box: two fresh child processes share cuda:0 and the process group is gloo
    (MESHVAE_DIST_BACKEND) because RCCL wants one device per rank.  Executes the multi-rank branch of bench.py end to end at
    the BENCHMARKED size -- the 5k model at 64 meshes per rank (BASELINE configs[2] per-rank shape): init, rank-0 broadcast,
    one all-reduce per step, barrier + MAX over ranks of the timed region, ONE JSON line from rank 0 only.  No scaling
    figure is expected from two ranks on one device."""
    import json
    import subprocess
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MESHVAE_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64",
                                       "--steps", "3", "--warmup", "1", "--prewarm-steps", "0", "--no-cpu-baseline",
                                       "--no-kernel-roofline"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines0 = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines0) == 1 and not outs[1][0].strip(), (outs[0][0][:300], outs[1][0][:300])   # rank 0 speaks, rank 1 does not
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["per_gpu_batch"] == 64
    assert d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak" and d["steps"] == 3
    assert d["final_loss"] == d["final_loss"] and 0 < d["final_loss"] < 1e6
    assert abs(d["value"] - 128 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "variants" not in d and "cpu_baseline" not in d           # single-GPU legs stay out of a multi-rank line


def test_step_beside_a_collective_stream_with_the_queue_budget_of_eight():
    """a data-parallel rank has a FOURTH busy hardware queue beside the step's three (the collective
    library's stream).  With GPU_MAX_HW_QUEUES=8 -- what bench.py sets and what TrainStep(group=...) asks for -- the step
    with a 1-rank RCCL all-reduce of the flat gradient buffer, and with a surrogate kernel stream ordered where the
    all-reduce goes, stays within 25 % of the plain step (measured: 0.470 / 0.503 against 0.459 ms,
    profiles/r05_fourth_queue.txt; under the default budget of four the RCCL form is 0.59-0.61 ms = +30 %).  Each mode in a
    fresh process (the budget is read when the HIP runtime starts); same loss in all three."""
    import re
    import subprocess
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", MASTER_PORT=str(_free_port()))
    got = {}
    for mode in ("plain", "surrogate", "rccl"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fourth_queue_probe.py"), "--mode", mode,
                              "--steps", "300"], env=env, capture_output=True, text=True, timeout=240)
        assert out.returncode == 0, out.stderr[-800:]
        m = re.search(r"GPU_MAX_HW_QUEUES=8: ([0-9.]+) ms/step .*loss ([0-9.]+)", out.stdout)
        assert m, out.stdout[-400:]
        got[mode] = (float(m.group(1)), m.group(2))
    print("[fourth queue, budget 8] " + " ".join(f"{k} {v[0]:.4f} ms" for k, v in got.items()))
    assert got["plain"][1] == got["surrogate"][1] == got["rccl"][1]
    assert got["surrogate"][0] <= 1.25 * got["plain"][0], got
    assert got["rccl"][0] <= 1.25 * got["plain"][0], got


def test_trainstep_warns_about_the_queue_budget_when_given_a_group(tmp_path):
    """... and TrainStep says so when it is handed a process group under a smaller budget (a RuntimeWarning, not an error:
    the step is correct either way, only slower)."""
    import subprocess
    code = (
        "import os, sys, warnings\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {PKG!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))\n"
        "import torch, torch.distributed as dist\n"
        "from conftest import TINY_CFG\n"
        "from model import load_topology\n"
        "from models.cheb_VAE import cheb_VAE\n"
        "from meshvae_hip.engine import TrainStep\n"
        "dev = torch.device('cuda:0'); torch.cuda.set_device(dev)\n"
        "dist.init_process_group('gloo', rank=0, world_size=1)\n"
        f"D, U, A, nn_ = load_topology(os.path.join({ROOT!r}, 'tests', 'golden', 'topology_tiny.npz'), dev)\n"
        "net = cheb_VAE(3, dict(TINY_CFG), D, U, A, nn_, model='optimal_sigma_VAE').to(dev).train()\n"
        "with warnings.catch_warnings(record=True) as w:\n"
        "    warnings.simplefilter('always')\n"
        "    TrainStep(net, 4, rehearse_allreduce=True)      # (a 1-rank group: the collective's stream without peers)\n"
        "print('WARNED' if any('GPU_MAX_HW_QUEUES' in str(x.message) for x in w) else 'SILENT')\n"
        "dist.destroy_process_group()\n")
    for budget, want in (("4", "WARNED"), ("8", "SILENT")):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=budget, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
        assert out.returncode == 0, out.stderr[-800:]
        assert want in out.stdout, (budget, out.stdout[-300:], out.stderr[-300:])
