#!/usr/bin/env python3
"""Headline benchmark: meshes/s, forward+backward(+all-reduce+Adam), 5k-vertex ChebConv VAE.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one train step of models.cheb_VAE on a synthetic batch of 64 meshes per GPU
(BASELINE.json configs[1]: default.cfg architecture on the 4998-vertex template, K=6, fp32,
dropout 0.2 on, x ~ N(0,1), x_gt = x as fp64 like main.py): forward, backward, one flat RCCL
all-reduce of the gradients when N > 1, fused Adam.  Inputs are resident in HBM before the
timed region.  Weak scaling: 64 meshes per rank.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "mesh-vae_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# The step runs on three HIP streams (main chain + two weight-gradient lanes); a process group adds RCCL's.
# The HIP runtime multiplexes streams onto 4 hardware queues by default, and with the extra streams the
# gradient lanes end up sharing the main chain's queue: measured 0.83 ms/step instead of 0.63 with a 1-rank
# RCCL group (tools/dist_overhead.sh, dist_overhead2.sh).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TOPOLOGY = os.path.join(ROOT, "tests", "golden", "topology_5k.npz")
CFG = {"n_layers": 4, "num_conv_filters": [16, 16, 16, 32, 32], "polygon_order": [6, 6, 6, 6, 6],
       "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}
# module-boundary HBM bytes per mesh, fp32, forward + backward
ALGO_BYTES_PER_MESH = 7.80e6
ALGO_FLOP_PER_MESH = 139.4e6
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def build_model(dev):
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    D, U, A, nn_ = load_topology(TOPOLOGY, dev)
    torch.manual_seed(666)
    return cheb_VAE(3, CFG, D, U, A, nn_, model="optimal_sigma_VAE").to(dev)


def time_kernel(fn, iters=30, warm=5):
    """Average device time (ms) of `fn` measured with HIP events on the launching stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_rooflines(net, B, dev):
    """Algorithmic bytes / measured duration of the level-0 (4998-vertex, 16-channel) kernels.
    Algorithmic bytes = every operand of the launch read or written exactly once;
    durations are HIP-event averages on the launching stream.  Each op below is one dominant
    kernel plus a <= 5 us helper launch (weight packing / partial-sum reduce), named in the key."""
    from meshvae_hip import check, lib
    from meshvae_hip.functional import workspace
    L = lib()
    net._prepare()
    lap = net._lap[0]
    N, C, K, Cout = net.num_nodes[0], 16, 6, 16
    st = torch.cuda.current_stream(dev).cuda_stream
    plane = B * N * C * 4                                   # bytes of one [B, N, 16] fp32 tensor
    x = torch.randn(B, N, C, device=dev)
    out = torch.empty(B, N, Cout, device=dev)
    signs = torch.empty(B, N, Cout // 4, dtype=torch.uint8, device=dev)
    dout = torch.randn(B, N, Cout, device=dev)
    W = torch.randn(K, C, Cout, device=dev) * 0.1
    bias = torch.zeros(Cout, device=dev)
    dW, db, dx = torch.empty_like(W), torch.empty_like(bias), torch.empty_like(x)
    ws_b = L.mvh_cheb_conv_bwd_ws_bytes(B, N, C, Cout, K)
    ws = workspace(ws_b, dev)
    res = {}
    sign_bytes = B * N * (Cout // 4)

    # the ops exactly as the train step runs them: fused ReLU with its signs kept as bytes
    def fwd():
        check(L.mvh_cheb_conv_fwd_signs(st, lap.fwd.ref, x.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                        signs.data_ptr(), B, N, C, Cout, K, ws.data_ptr(), ws_b))
    res["k_cheb_lds<16,5,1024,4,false> (+k_pack_w): conv+relu fwd L0 16->16"] = dict(
        ms=time_kernel(fwd), bytes=2 * plane + sign_bytes, launches_per_step=1)      # read x; write out, signs

    def bwd_dw():
        check(L.mvh_cheb_conv_bwd_signs(st, lap.fwd.ref, lap.bwd.ref, x.data_ptr(), W.data_ptr(), out.data_ptr(),
                                        signs.data_ptr(), dout.data_ptr(), None, dW.data_ptr(), db.data_ptr(),
                                        B, N, C, Cout, K, ws.data_ptr(), ws_b))
    res["k_cheb_dw_lds<16,10,512,4> (+k_dw_reduce): conv dW/db L0 16->16"] = dict(
        ms=time_kernel(bwd_dw), bytes=2 * plane + sign_bytes, launches_per_step=1)   # read x, dout, relu signs

    def bwd_dx():
        check(L.mvh_cheb_conv_bwd_signs(st, lap.fwd.ref, lap.bwd.ref, x.data_ptr(), W.data_ptr(), out.data_ptr(),
                                        signs.data_ptr(), dout.data_ptr(), dx.data_ptr(), None, None,
                                        B, N, C, Cout, K, ws.data_ptr(), ws_b))
    res["k_cheb_lds<16,5,1024,4,true> (+k_pack_w): conv dX L0 16->16"] = dict(
        ms=time_kernel(bwd_dx), bytes=2 * plane + sign_bytes, launches_per_step=1)   # read dout, signs; write dx
    return res


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the named kernel from the committed PMC passes (profiles/): rocprofv3
    cannot be driven from inside this process, so the counters are collected separately and read here."""
    path = os.path.join(ROOT, "profiles", "r01_j_pmc_traffic.json")
    try:
        table = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    for k, v in table.items():
        if kernel_key.replace(" ", "").startswith(k.replace(" ", "")):
            return v.get("hbm_bytes")
    return None


def host_cores():
    """CPUs this process may actually use: min(affinity, cgroup quota) -- the GPU box exposes
    256 hardware threads but caps the container at 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(B, steps):
    """The CPU oracle (a port of the reference dataflow) timed on this box's host cores."""
    from oracle import cheb_oracle as O
    n_threads = host_cores()
    torch.set_num_threads(n_threads)
    topo = O.Topology(np.load(TOPOLOGY))
    torch.manual_seed(666)
    sd = O.init_state_dict(CFG, topo)
    net = O.OracleVAE(CFG, topo, sd, requires_grad=True)
    net.training = True
    x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(0))
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2)

    def step():
        for p in net.p.values():
            p.grad = None
        net.forward(x, x.double(), y, "train")[0].backward()
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "meshes/s", "cores": n_threads, "kind": "port",
            "sample": f"{steps} train steps (fwd+bwd) of B={B} on the same 5k model after 1 warm-up step, "
                      f"torch {torch.__version__} CPU, {n_threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="meshes per GPU")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (default: eager C++ launch sequence -- on ROCm 7.2 the "
                         "graph executor serialises the side-stream branch, eager overlaps it and is ~15%% faster)")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--micro", type=int, default=1, help="independent chains the per-GPU batch is pipelined over")
    ap.add_argument("--prewarm-steps", type=int, default=600,
                    help="untimed steps run BEFORE the --warmup steps (clock / power-state ramp of a cold GPU: the "
                         "first process on a fresh box was seen 25 %% slow otherwise).  A step COUNT, not a duration: "
                         "every rank must issue the same number of gradient all-reduces.  0 disables")
    ap.add_argument("--rehearse-allreduce", action="store_true",
                    help="1 GPU: bring up a 1-rank RCCL group and run the gradient collective anyway (rehearsal of the "
                         "multi-GPU stream hand-over on a one-GPU box)")
    ap.add_argument("--ar-overlap", action="store_true", help="two-bucket overlapped gradient all-reduce (opt-in)")
    ap.add_argument("--seed", type=int, default=666, help="weights: this seed on every rank; noise: seed + rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE line (rank 0's JSON): libraries that print there (RCCL writes a five-line version
    # banner to stdout when the first communicator comes up) are pointed at stderr until that line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = args.rehearse_allreduce                             # 1-rank rehearsal of the RCCL path
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from meshvae_hip.engine import TrainStep
    net = build_model(dev)
    net.train()
    B = args.batch
    # weights: same seed on every rank AND a broadcast from rank 0 inside TrainStep; reparameterisation noise and
    # dropout masks: private generators seeded seed + rank, so no two ranks draw the same noise for their shards
    step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=bool(args.graph), m_type="train",
                     n_micro=args.micro, noise_seed=args.seed, rehearse_allreduce=args.rehearse_allreduce,
                     overlap_allreduce=args.ar_overlap)
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, 4998, 3, generator=g)
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2)
    step.x.copy_(x)
    step.x_gt = x.double().to(dev)                   # fp64 ground truth, as main.py:69-70 hands it over
    step.y.copy_(y)
    if step.use_graph:
        step.capture()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(max(args.prewarm_steps, 0)):   # not part of the contract's W warm-up steps: extra untimed work
        step.step()
        if i % 50 == 49:
            torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.step()
    barrier()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss = float(step.out[0])
    assert np.isfinite(loss), "non-finite loss in the benchmark step"

    if rank == 0:
        meshes_per_s = world * B * args.steps / dt
        out = {
            "metric": "meshes/sec fwd+bwd, 5k-vertex ChebConv VAE",
            "value": meshes_per_s, "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": max(args.prewarm_steps, 0),
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: default.cfg 5k-vertex K=6 ChebConv VAE train step "
                                   "(fwd+bwd+grad all-reduce+Adam), 64 meshes/GPU, fp32, dropout 0.2",
                       "global_batch": world * B, "per_gpu_batch": B, "vertices": 4998,
                       "parallelism": f"dp{world}", "hipgraph": bool(step.use_graph),
                       "micro_batches": step.n_micro,
                       "precision": "fp32 storage and arithmetic: configs[1] names bf16 storage, this run keeps the "
                                    "reference's fp32 (the higher precision, and the one the 1e-4 parity bar is stated in)"},
            "step_roofline": {"bound": "hbm", "achieved": meshes_per_s / world * ALGO_BYTES_PER_MESH / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": meshes_per_s / world * ALGO_BYTES_PER_MESH / 1e9 / HBM_PEAK_GBS,
                              "note": "whole step per GPU: meshes/s x 7.80 MB/mesh (SURVEY 8(d))"},
            "final_loss": loss,
        }
        if not args.no_kernel_roofline:
            ks = kernel_rooflines(net, B, dev)
            name = max((k for k in ks if k.startswith("k_")), key=lambda k: ks[k]["ms"] * ks[k]["launches_per_step"])
            # (the dominant kernel of the step by total time: see profiles/ for the rocprofv3 view)
            d = ks[name]
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(name),
                               "avg_launch_us": d["ms"] * 1e3, "algorithmic_bytes_per_launch": d["bytes"]}
            out["kernels"] = {k: {"avg_ms": v["ms"], "algo_GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for k, v in ks.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, 8)   # ~10-20 s of CPU work on 16 cores
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
