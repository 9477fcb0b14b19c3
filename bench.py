#!/usr/bin/env python3
"""Headline benchmark: meshes/s, forward+backward(+all-reduce+Adam), 5k-vertex ChebConv VAE.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default (`--config train5k`): one "step" = one train step of models.cheb_VAE on a synthetic batch of 64 meshes
per GPU (BASELINE.json configs[1]: default.cfg architecture on the 4998-vertex template, K=6, dropout 0.2 on,
x ~ N(0,1), x_gt = x as fp64 like main.py): forward, backward, one flat RCCL all-reduce of the gradients when
N > 1, fused Adam.  Inputs are resident in HBM before the timed region.  Weak scaling: 64 meshes per rank.
Prints ONE JSON line on rank 0.

Other BASELINE configurations, same JSON schema (1 GPU):
    --config hires20k   configs[3]: 19 992-vertex template, 6 levels, K = 10, train step, 31.2 MB/mesh bound
    --config infer      configs[4]: crecon.py:170-192 inference path (encode -> classify -> z_mean -> 2 x decode),
                        hipGraph-captured, latency at batch 1 / 32 / 256 (value = the B = 32 latency)
    --dtype bf16        configs[1] as worded: activations stored bf16 in HBM, fp32 accumulation (3.90 MB/mesh bound)
"""
import argparse
import csv
import glob
import hashlib
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
# stream value waits of the asynchronous launcher on the command processor, not as a spinning shader (meshvae_hip/__init__.py);
# single-process runs only (the launcher serves variants.reference_loop; a multi-rank line never runs it)
if os.environ.get("WORLD_SIZE", "1") in ("", "1"):
    os.environ.setdefault("GPU_STREAMOPS_CP_WAIT", "1")
    os.environ.setdefault("MESHVAE_ASYNC", "1")          # (opt-in of the module path)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CFG_5K = {"n_layers": 4, "num_conv_filters": [16, 16, 16, 32, 32], "polygon_order": [6, 6, 6, 6, 6],
          "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}
# BASELINE configs[3] as SURVEY 8(d) pins it: 1->4 subdivision of the 5k template, 6 levels, K = 10
CFG_20K = {"n_layers": 5, "num_conv_filters": [16, 16, 16, 32, 32, 32], "polygon_order": [10] * 6,
           "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}
CFG = CFG_5K                                  # (older tools import bench.CFG / bench.TOPOLOGY)
TOPOLOGY = os.path.join(GOLDEN, "topology_5k.npz")
# module-boundary HBM bytes per mesh, forward + backward
ALGO_BYTES_PER_MESH = {("train5k", "f32"): 7.80e6, ("train5k", "bf16"): 3.90e6, ("hires20k", "f32"): 31.2e6,
                       ("hires20k", "bf16"): 15.6e6}
ALGO_BYTES_FWD_PER_MESH = 2.771e6             # forward only (one encoder + one decoder pass), fp32
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
WORKLOADS = {
    "train5k": "configs[1]: default.cfg 5k-vertex K=6 ChebConv VAE train step (fwd+bwd+grad all-reduce+Adam), "
               "dropout 0.2",
    "hires20k": "configs[3]: 19 992-vertex template (1->4 subdivision of the 5k one), 6 levels, K=10 ChebConv VAE train "
                "step (fwd+bwd+Adam), dropout 0.2",
    "infer": "configs[4]: inference path of crecon.py:170-192 on the 5k template (encoder -> classifier -> z_mean -> "
             "decoder for the predicted and the opposite label), no_grad, hipGraph replay, latency at batch 1/32/256",
}


def build_model(dev, config="train5k"):
    from model import load_topology
    from models.cheb_VAE import cheb_VAE
    hires = config == "hires20k"
    D, U, A, nn_ = load_topology(os.path.join(GOLDEN, "topology_20k.npz" if hires else "topology_5k.npz"), dev)
    torch.manual_seed(666)
    return cheb_VAE(3, dict(CFG_20K if hires else CFG_5K), D, U, A, nn_, model="optimal_sigma_VAE").to(dev)


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


# ----------------------------------------------------------------------------------------------------------------------
# per-kernel roofline: every ChebConv launch of the model, timed live in isolation through the C ABI
def source_sha():
    """Hash of the kernel sources a profile under profiles/ was taken on (tools/pmc_fold.py writes the same hash
    into <tag>_pmc.json): profile figures are only quoted when they describe the kernels being benchmarked."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mesh-vae_amd", "csrc", "*.h*")) +
                    [os.path.join(ROOT, "include", "meshvae_hip.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def matching_profile(config="train5k", dtype="f32"):
    """(tag, pmc table, {kernel: (calls, total_us, avg_us)}) of the newest profiles/*_pmc.json taken on these
    sources with this bench configuration, or (None, {}, {})."""
    sha = source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("_src_sha") != sha or d.get("_config", "train5k") != config or d.get("_dtype", "f32") != dtype:
            continue
        tag = d.get("_tag") or os.path.basename(path)[:-len("_pmc.json")]
        stats = {}
        try:
            for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))):
                if r["kernel"] != "TOTAL":
                    stats[r["kernel"]] = (int(r["calls"]), float(r["total_us"]), float(r["avg_us"]))
        except (OSError, KeyError, ValueError):
            pass
        return tag, d.get("kernels", {}), stats
    return None, {}, {}


def patch_plan_of(lap):
    """The mvh_patch_plan_t a Laplacian operator (or a CsrStruct pair standing in for one) carries, or None."""
    import ctypes
    from meshvae_hip import PatchPlanStruct, lib
    ptr = lap.fwd.struct.patch
    if not ptr or lib().mvh_debug_get(b"no_patch"):
        return None
    return ctypes.cast(ptr, ctypes.POINTER(PatchPlanStruct)).contents


class _LapWithPlan:
    """The step engine's view of a level: the Laplacian's descriptors with the plan that carries the pooling rows of the
    level's un-pooling operator (engine.NativeStep) -- the kernel instance the TRAIN STEP launches, for the isolated timing."""

    class _Side:
        def __init__(self, struct, ell_pairs):
            import ctypes
            self.struct, self.ell_pairs, self.ref = struct, ell_pairs, ctypes.byref(struct)

    def __init__(self, lap, plan_struct):
        import ctypes
        from meshvae_hip import CsrStruct
        f, b = CsrStruct.from_buffer_copy(lap.fwd.struct), CsrStruct.from_buffer_copy(lap.bwd.struct)
        f.patch = b.patch = ctypes.addressof(plan_struct)
        self.fwd, self.bwd = self._Side(f, lap.fwd.ell_pairs), self._Side(b, lap.bwd.ell_pairs)


def lds_kernel_name(kind, lap, N, Cin, Cout, half=False):
    """Template instance the launchers of csrc/cheb_lds.hip / cheb_dw_lds.hip (bf16 rows at the 5k level: cheb_l0h.hip /
    cheb_dw_l0h.hip) pick for a layer (mirrors try_cheb_lds / try_cheb_dw_lds), or None when the layer takes another
    path (split path, stack pipeline)."""
    pw = 8 if lap.fwd.ell_pairs > 4 else 4
    pp = patch_plan_of(lap)
    # (bf16 storage: the backward only -- the forward stays on the bf16 matrix-pipe kernel k_cheb_l0h)
    if pp is not None and Cin == 16 and Cout == 16 and (kind == "dX+dW" or (kind == "fwd" and not half)):
        # vertex-patch kernels (csrc/cheb_patch.hip: FwdCfg / BwdCfg; the SU variant needs that many core tiles per wave)
        tiles = pp.min_core // 16
        if kind == "fwd":
            return f"k_patch_fwd<1024,7,6,{4 if tiles // 16 >= 4 else 0}>"
        return f"k_patch_bwd<8,8,14,11,40,{10 if tiles // 8 >= 10 else 0}>"
    if kind == "dX+dW":
        return None                                                      # (only the patch kernels fuse the two gradients)
    if half and Cin == 16 and Cout == 16 and 2048 < N + 1 <= 5120 and lap.fwd.ell_pairs <= 4 and \
            not (0 < lap.fwd.struct.n_active and 4 * lap.fwd.struct.n_active <= N):
        return "k_cheb_dw_l0h" if kind == "dW" else f"k_cheb_l0h<{'true' if kind == 'dX' else 'false'}>"
    if N + 1 > 5120 or lap.fwd.ell_pairs <= 0:
        return None
    if 0 < lap.fwd.struct.n_active and 4 * lap.fwd.struct.n_active <= N:
        return None                                                      # mostly-isolated Laplacian: split path

    def shape(cq, dw):
        if N + 1 <= 1024:
            return 1, 0
        if N + 1 <= 2048:
            return 2, 0
        if cq <= 16:
            return (10, 512) if dw else (5, 1024)
        return None
    if kind == "dW":
        cp, cq = (Cin, Cout) if Cin <= Cout else (Cout, Cin)
        if cq not in (8, 16, 32):
            return None
        s = shape(cq, True)
        if s is None:
            return None
        if (cp + 3) // 4 == 1:
            cq = 4                                                       # Q-split
        return f"k_cheb_dw_lds<{cq},{s[0]},{s[1]},{pw}>"
    cq = Cin if kind == "fwd" else Cout
    if cq not in (3, 8, 16, 32):
        return None
    s = shape(cq, False)
    if s is None:
        return None
    return f"k_cheb_lds<{cq},{s[0]},{s[1]},{pw},{'true' if kind == 'dX' else 'false'}>"


def conv_ops(net, B, dev, dtype="f32"):
    """One entry per (conv layer, fwd | dX | dW) of the model: a closure that launches it through the C ABI as the
    public fused-ReLU ops do, its algorithmic bytes (every operand read or written exactly once: DESIGN.md 4) and
    the name of the kernel instance that dominates it.  dtype "bf16": layers whose channel counts are multiples of 4
    run the bf16-storage ops (mvh_cheb_conv_*_bf16); the 3-channel first / final layers keep fp32 tensors here (in
    the step only their 16-channel side is bf16)."""
    from meshvae_hip import check, lib
    L = lib()
    net._prepare()
    n = net.n_layers
    f = net.filters
    layers = []                                  # (label, lap, N, Cin, Cout, K, relu, has_dx)
    for i in range(n):
        layers.append((f"enc{i}", net._lap[i], net.num_nodes[i], f[i], f[i + 1], net.K[i], True, i > 0))
    for i in range(n):
        lvl = n - i - 1
        layers.append((f"dec{i}", net._lap[lvl], net.num_nodes[lvl], f[n + 1 - i], f[n - i], net.K[i], True, True))
    layers.append(("final", net._lap_final, net.num_nodes[0], f[1], f[0], net.K[n], False, True))
    st = torch.cuda.current_stream(dev).cuda_stream
    ops = []
    ws_b = max(max(L.mvh_cheb_conv_ws_bytes(B, N, Cin, Cout, K), L.mvh_cheb_conv_bwd_ws_bytes(B, N, Cin, Cout, K))
               for _, _, N, Cin, Cout, K, _, _ in layers)
    ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)   # one scratch buffer: the ops run one after the other
    keep_plans = []
    for label, lap, N, Cin, Cout, K, relu, has_dx in layers:
        half = dtype == "bf16" and Cin % 4 == 0 and Cout % 4 == 0
        if label.startswith("dec") and Cin == 16 and Cout == 16 and patch_plan_of(lap) is not None:
            from meshvae_hip import topology
            lvl = [i for i in range(n) if net._lap[i] is lap][0]
            got = topology.patch_plan(lap, int(K) - 1, net._up[lvl])
            if got is not None:
                keep_plans.append(got)
                lap = _LapWithPlan(lap, got[0])
        td = torch.bfloat16 if half else torch.float32
        esize = 2 if half else 4
        x = torch.randn(B, N, Cin, device=dev).to(td)
        out = torch.empty(B, N, Cout, device=dev, dtype=td)
        dout = torch.randn(B, N, Cout, device=dev).to(td)
        W = torch.randn(K, Cin, Cout, device=dev) * 0.1
        bias = torch.zeros(Cout, device=dev) if relu else None
        dW, dx = torch.empty_like(W), torch.empty_like(x)
        act = 1 if relu else 0
        db = torch.empty(Cout, device=dev) if relu else None
        use_signs = relu and Cout % 4 == 0 and K > 1
        signs = torch.empty(B, N, max(Cout // 4, 1), dtype=torch.uint8, device=dev)
        # levels of the streaming kernels (> 5119 vertices, fp32): the step's forward SAVES the T_k stack for the weight
        # gradient (vae_step.hip, txEnc / txDec); the isolated ops do the same through the tx_saved argument of the plain
        # entry points, so that "conv dW" is timed without rebuilding the stack
        tx = None
        if N + 1 > 5120 and not half and K > 1 and label != "final":   # (the final layer takes the split path: no stack)
            tx = torch.empty((K - 1) * B * N * Cin, device=dev)
            use_signs = False
        keep = (x, out, dout, W, bias, dW, dx, db, signs, ws, tx)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731

        def fwd(lap=lap, x=x, W=W, bias=bias, out=out, signs=signs, N=N, Cin=Cin, Cout=Cout, K=K, ws=ws, ws_b=ws_b,
                use_signs=use_signs, relu=relu, half=half, act=act, tx=tx):
            if half:
                check(L.mvh_cheb_conv_fwd_bf16(st, lap.fwd.ref, p(x), p(W), p(bias), p(out), p(signs) if relu else None, B, N,
                                               Cin, Cout, K, act, p(ws), ws_b))
            elif use_signs:
                check(L.mvh_cheb_conv_fwd_signs(st, lap.fwd.ref, p(x), p(W), p(bias), p(out), p(signs), B, N, Cin, Cout,
                                                K, p(ws), ws_b))
            else:
                check(L.mvh_cheb_conv_fwd(st, lap.fwd.ref, p(x), p(W), p(bias), p(out), p(tx), B, N, Cin, Cout, K,
                                          int(relu), p(ws), ws_b))

        def bwd(want_dx, want_dw, lap=lap, x=x, W=W, out=out, signs=signs, dout=dout, dx=dx, dW=dW, db=db, N=N, Cin=Cin,
                Cout=Cout, K=K, ws=ws, ws_b=ws_b, use_signs=use_signs, relu=relu, half=half, act=act, tx=tx):
            a_dx, a_dw, a_db = (p(dx) if want_dx else None), (p(dW) if want_dw else None), (p(db) if want_dw else None)
            if half:
                check(L.mvh_cheb_conv_bwd_bf16(st, lap.fwd.ref, lap.bwd.ref, p(x), p(W), p(signs) if relu else None, p(dout),
                                               a_dx, a_dw, a_db, B, N, Cin, Cout, K, act, p(ws), ws_b))
            elif use_signs:
                check(L.mvh_cheb_conv_bwd_signs(st, lap.fwd.ref, lap.bwd.ref, p(x), p(W), p(out), p(signs), p(dout), a_dx,
                                                a_dw, a_db, B, N, Cin, Cout, K, p(ws), ws_b))
            else:
                check(L.mvh_cheb_conv_bwd(st, lap.fwd.ref, lap.bwd.ref, p(x), p(W), p(out), p(dout), p(tx), a_dx, a_dw,
                                          a_db, B, N, Cin, Cout, K, int(relu), p(ws), ws_b))
        fwd()                                    # forward once: valid `out` / signs for the backward closures
        pin, pout = B * N * Cin * esize, B * N * Cout * esize
        sb = B * N * (Cout // 4) if (use_signs or (half and relu)) else (pout if relu else 0)   # ReLU mask: sign bytes, else the fp32 output
        desc = f"{label} N={N} {Cin}->{Cout} K={K}" + (" bf16" if half else "")
        ops.append(dict(op=f"conv fwd {desc}", kernel=lds_kernel_name("fwd", lap, N, Cin, Cout, half), fn=fwd,
                        bytes=pin + pout + (B * N * (Cout // 4) if (use_signs or (half and relu)) else 0), keep=keep))
        if lds_kernel_name("dX+dW", lap, N, Cin, Cout, half) is not None:
            # vertex-patch level: both gradients come out of ONE launch (minimum traffic: x, dout and the sign bytes read once, dx written)
            ops.append(dict(op=f"conv dX+dW {desc}", kernel=lds_kernel_name("dX+dW", lap, N, Cin, Cout, half),
                            fn=lambda bwd=bwd: bwd(True, True), bytes=2 * pin + pout + sb, keep=keep + (keep_plans,)))
            continue
        if has_dx:
            ops.append(dict(op=f"conv dX {desc}", kernel=lds_kernel_name("dX", lap, N, Cin, Cout, half),
                            fn=lambda bwd=bwd: bwd(True, False), bytes=pout + sb + pin, keep=keep))
        if label == "enc0" and lds_kernel_name("fwd", lap, N, Cin, Cout) is not None:
            # the step takes this layer's weight gradient from a saved Chebyshev stack at the pooled rows
            # (k_cheb_tstack + k_stack_dw, csrc/cheb_tstack.hip), which the public per-layer op cannot express
            continue
        ops.append(dict(op=f"conv dW {desc}", kernel=lds_kernel_name("dW", lap, N, Cin, Cout, half),
                        fn=lambda bwd=bwd: bwd(False, True), bytes=pin + pout + sb, keep=keep))
    return ops


def time_kernel_median(fn, repeats=5, iters=30, warm=5):
    """Median of `repeats` HIP-event averages (time_kernel): one slow repeat -- another process's burst on the box, a
    clock dip -- must not decide which launch the bench calls dominant (BENCH_r03: `conv fwd dec0` once 43 us for 10)."""
    ts = sorted(time_kernel(fn, iters, warm if r == 0 else 1) for r in range(repeats))
    return ts[len(ts) // 2]


def step_launches(kernel, B, config, dtype):
    """How many launches of `kernel` ONE train step issues (engine rule, csrc/vae_step.hip `l0_split`): the 5k level's
    weight-gradient kernel runs as `l0_lane` part-batch launches on the dense lane for 56 < B <= 64 (bf16 storage: only with
    the l0_lane_bf switch), every other conv kernel once."""
    from meshvae_hip import lib
    L = lib()
    if config != "train5k" or not (kernel or "").startswith(("k_cheb_dw_lds<16,10,512,4>", "k_cheb_dw_l0h")):
        return 1
    lane = L.mvh_debug_get(b"l0_lane")
    fits = (56 < B <= 64) or (L.mvh_debug_get(b"l0_lane_any") and B >= 16)
    if lane > 1 and fits and (dtype == "f32" or L.mvh_debug_get(b"l0_lane_bf")):
        return int(lane)
    return 1


def kernel_roofline(net, B, dev, config="train5k", dtype="f32", kinds=("fwd", "dX", "dW", "dX+dW")):
    """The `roofline` object of the bench line, for the conv launch that costs most.

    Two figures, both reproducible from what the line and profiles/ carry:
      * `frac` (isolated): algorithmic bytes of the whole-batch launch / its live HIP-event time (median of 5 repeats of a
        30-launch average, on the launching stream) / peak.  This is the kernel's own speed.
      * `in_step.frac`: algorithmic bytes of ONE launch as the train step issues it / the rocprofv3 `avg_us` of that kernel
        in the committed profile taken on these sources (profiles/<tag>_kernel_stats.csv) / peak -- lower, because in the
        step the launch shares the chip with the other two streams; `in_step.traffic_ratio` = PMC HBM bytes per launch
        (profiles/<tag>_pmc.json, hbm_bytes) / algorithmic bytes per launch.  None when no committed profile matches the
        source hash of the kernels being benchmarked."""
    ops = [o for o in conv_ops(net, B, dev, dtype) if o["op"].split()[1] in kinds]
    for o in ops:
        o["ms"] = time_kernel_median(o["fn"])
    tag, pmc, stats = matching_profile(config, dtype)
    # Ranking: every conv launch of the model by its ISOLATED HIP-event time (all 25, not a hand-picked few).  The
    # rocprofv3 totals of the committed profile are listed beside it (rocprof_top): under the step's three concurrent
    # streams a small kernel's rocprof duration includes the time its workgroups wait for CUs that a chip-filling
    # level-0 kernel of another stream holds, so the rocprof total ranks waiting, not work; both views are printed.
    ranking = "isolated HIP-event time of every conv launch, median of 5 x 30-launch averages (x 1 launch per step each)"
    top = max(ops, key=lambda o: o["ms"])
    rocprof_top = [{"kernel": k, "calls": v[0], "total_us": v[1], "avg_us": v[2]}
                   for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:5]]
    ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
    prof = pmc.get(top["kernel"] or "", {})
    n_launch = step_launches(top["kernel"], B, config, dtype)
    pmc_bytes = prof.get("hbm_bytes")                      # per launch AS PROFILED (i.e. per part-batch launch in the step)
    out = {"bound": "hbm", "kernel": f'{top["kernel"] or "split/stack path"}: {top["op"]}', "achieved": ach,
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": None if pmc_bytes is None else n_launch * pmc_bytes,   # per whole-batch launch, like `achieved`
           "avg_launch_us": top["ms"] * 1e3, "algorithmic_bytes_per_launch": top["bytes"], "ranking": ranking}
    st = stats.get(top["kernel"] or "")
    in_bytes = top["bytes"] // n_launch
    in_step = {"launches_per_step": n_launch, "meshes_per_launch": B // n_launch, "algorithmic_bytes_per_launch": in_bytes,
               "avg_us": st[2] if st else None, "achieved": None, "frac": None,
               "pmc_hbm_bytes_per_launch": pmc_bytes, "traffic_ratio": None if not pmc_bytes else pmc_bytes / in_bytes,
               "kernel_stats": f"profiles/{tag}_kernel_stats.csv" if tag else None,
               "pmc": f"profiles/{tag}_pmc.json" if tag else None,
               "note": "the launch as the train step issues it, beside the other two streams: algorithmic_bytes_per_launch / "
                       "avg_us (rocprofv3 --kernel-trace --stats of the committed profile taken on these sources) / peak; "
                       "traffic_ratio = hbm_bytes of the PMC passes ((2 FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction of "
                       "MI355X_MICROARCH.md) / algorithmic bytes.  None: no committed profile matches these kernel sources"}
    if st and st[2] > 0:
        in_step["achieved"] = in_bytes / (st[2] * 1e-6) / 1e9
        in_step["frac"] = in_step["achieved"] / HBM_PEAK_GBS
    out["in_step"] = in_step
    if rocprof_top:
        out["rocprof_top"] = {"file": f"profiles/{tag}_kernel_stats.csv", "by_total_time": rocprof_top}
    if prof:
        out["profile"] = {"tag": tag, "mfma_util": prof.get("mfma_util"), "lds_conflict_frac": prof.get("lds_conflict_frac"),
                          "scratch_bytes_per_lane": prof.get("scratch_bytes_per_lane"),
                          "rocprof_avg_us": st[2] if st else None}
    table = {o["op"]: {"kernel": o["kernel"], "avg_us": round(o["ms"] * 1e3, 2),
                       "algo_GBps": round(o["bytes"] / (o["ms"] * 1e-3) / 1e9, 1)} for o in ops}
    return out, table


# ----------------------------------------------------------------------------------------------------------------------
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


def cpu_baseline(config, B, budget_s=20.0):
    """The CPU oracle (a port of the reference dataflow) timed on this box's host cores, on a bounded sample of the
    same workload: steps are repeated until ~budget_s seconds of CPU work have been spent (at least 2)."""
    from oracle import cheb_oracle as O
    n_threads = host_cores()
    torch.set_num_threads(n_threads)
    hires = config == "hires20k"
    cfg = CFG_20K if hires else CFG_5K
    topo = O.Topology(np.load(os.path.join(GOLDEN, "topology_20k.npz" if hires else "topology_5k.npz")))
    torch.manual_seed(666)
    sd = O.init_state_dict(cfg, topo)
    train = config != "infer"
    net = O.OracleVAE(cfg, topo, sd, requires_grad=train)
    net.training = train
    n0 = int(topo.num_nodes[0])
    x = torch.randn(B, n0, 3, generator=torch.Generator().manual_seed(0))
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2)

    def step():
        if train:
            for p in net.p.values():
                p.grad = None
            net.forward(x, x.double(), y, "train")[0].backward()
        else:                                    # the forward passes of crecon.py:170-192: 1 encoder + 2 decoders
            with torch.no_grad():
                net.forward(x, x, y, "test")     # encoder + heads + decoder + loss
                net.forward(x, x, 1 - y, "test")  # (second decode; its encoder pass makes the sample an upper bound)
    step()
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 64):
        step()
        n += 1
    dt = time.perf_counter() - t0
    what = "train steps (fwd+bwd)" if train else "inference passes (2 x full eval forward: >= encode + 2 x decode)"
    return {"value": B * n / dt, "unit": "meshes/s", "cores": n_threads, "kind": "port",
            "sample": f"{n} {what} of B={B} on the same model after 1 warm-up, torch {torch.__version__} CPU, "
                      f"{n_threads} threads, {dt:.1f} s"}


# ----------------------------------------------------------------------------------------------------------------------
def estimate_diff_fn(net):
    """The VAE half of crecon.estimate_diff (crecon.py:170-192) over the drop-in module API."""
    def run(x):
        h = net.encoder(x)
        y_hat = net.classifier(h)
        y = torch.nn.functional.one_hot(y_hat.argmax(-1), 2).to(torch.float32)
        mu = net.z_mean(torch.cat([y, h], -1))
        return net.sample(y, mu), net.sample(1.0 - y, mu), y_hat
    return run


def capture_inference(fn, x, dev):
    """hipGraph of fn(x) (static input buffer x): -> (graph, outputs of the captured call)."""
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):                # warm the workspaces of the capture stream
        for _ in range(3):
            fn(x)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(x)
    return g, out


def estimate_diff_batched_fn(net):
    """The same inference with the two decodes (own label, opposite label) as ONE 2B-mesh decoder pass -- what the
    package's own crecon_ops.estimate_diff_device does; the decoder is per-mesh, so the results are bitwise the two
    calls' (asserted by the caller)."""
    def run(x):
        h = net.encoder(x)
        y_hat = net.classifier(h)
        y = torch.nn.functional.one_hot(y_hat.argmax(-1), 2).to(torch.float32)
        mu = net.z_mean(torch.cat([y, h], -1))
        both = net.sample(torch.cat([y, 1.0 - y]), torch.cat([mu, mu]))
        return both[:x.shape[0]], both[x.shape[0]:], y_hat
    return run


def infer_latencies(dev, steps, warmup, batched=False):
    """(hipGraph replay, eager) latency in ms of encode + classify + 2 x decode at B = 1 / 32 / 256; the replay is checked
    bitwise against the eager result.  batched: the two decodes as one 2B-mesh pass (checked bitwise against the two
    calls)."""
    net = build_model(dev).eval()
    fn = estimate_diff_batched_fn(net) if batched else estimate_diff_fn(net)
    lat, eager = {}, {}
    iters = max(steps, 20)
    for B in (1, 32, 256):
        x = torch.randn(B, 4998, 3, device=dev)
        if batched:
            with torch.no_grad():
                for a, b in zip(fn(x), estimate_diff_fn(net)(x)):
                    assert torch.equal(a, b), "batched decode differs from the two decoder calls"
        with torch.no_grad():
            for _ in range(max(warmup, 3)):
                ref = fn(x)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                fn(x)
            torch.cuda.synchronize(dev)
            eager[B] = (time.perf_counter() - t0) / iters * 1e3
            g, out = capture_inference(fn, x, dev)
            g.replay()
            torch.cuda.synchronize(dev)
            for a, b in zip(out, ref):
                assert torch.equal(a, b), "hipGraph replay differs from the eager result"
            for _ in range(max(warmup, 3)):
                g.replay()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize(dev)
            lat[B] = (time.perf_counter() - t0) / iters * 1e3
    return net, lat, eager, iters


def run_infer(args, dev, emit):
    """BASELINE configs[4]: hipGraph-captured inference latency at B = 1 / 32 / 256 through the reference-API module."""
    net, lat, eager, iters = infer_latencies(dev, args.steps, args.warmup)
    out = {"metric": "inference latency (encode + classify + 2 x decode, hipGraph replay), 5k-vertex ChebConv VAE, batch 32",
           "value": lat[32], "unit": "ms", "n_gpus": 1, "steps": iters, "warmup": max(args.warmup, 3),
           "ms_per_step": lat[32], "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "config": {"workload": WORKLOADS["infer"], "vertices": 4998, "hipgraph": True,
                                           "replay_equals_eager_bitwise": True},
           "latency_ms": {f"b{b}": lat[b] for b in lat}, "eager_latency_ms": {f"b{b}": eager[b] for b in eager},
           "meshes_per_s": {f"b{b}": b / lat[b] * 1e3 for b in lat}}
    # the package's own op for the same inference (crecon_ops.estimate_diff_device: both decodes in one 2B-mesh pass);
    # never `value` -- the headline stays the reference script's call sequence (two net.sample calls)
    _, lat2, eager2, _ = infer_latencies(dev, args.steps, args.warmup, batched=True)
    out["batched_decode"] = {"latency_ms": {f"b{b}": lat2[b] for b in lat2}, "eager_latency_ms": {f"b{b}": eager2[b] for b in eager2},
                             "equals_two_calls_bitwise": True,
                             "note": "own label and opposite label decoded in one 2B-mesh pass (crecon_ops.estimate_diff_device)"}
    # one inference = 1 encoder + 2 decoder passes: ~1.5 x the forward's module-boundary bytes
    bytes_per_mesh = 1.5 * ALGO_BYTES_FWD_PER_MESH
    ach = 256 / lat[256] * 1e3 * bytes_per_mesh / 1e9
    out["step_roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                            "note": "B = 256 replay: meshes/s x 1.5 x 2.77 MB/mesh (encoder + two decoder passes)"}
    if not args.no_kernel_roofline:
        out["roofline"], out["kernels"] = kernel_roofline(net, 256, dev, "infer", kinds=("fwd",))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline("infer", 32)
    emit(out)


def timed_variant(dev, config, dtype, B, steps, warmup, prewarm, seed):
    """One more driver-timed train configuration with the main leg's protocol (same TrainStep, synthetic batch resident
    in HBM, W untimed + K timed steps between device synchronisations) -> the figures of its own bench line."""
    from meshvae_hip.engine import TrainStep
    net = build_model(dev, config).train()
    step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=False, m_type="train", noise_seed=seed, storage=dtype)
    x = torch.randn(B, net.num_nodes[0], 3, generator=torch.Generator().manual_seed(0))
    step.x.copy_(x)
    step.x_gt = x.double().to(dev)
    step.y.copy_(torch.nn.functional.one_hot(torch.arange(B) % 2, 2))
    for i in range(prewarm + warmup):
        step.step()
        if i % 50 == 49:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step.step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    loss = float(step.out[0])
    assert np.isfinite(loss), f"non-finite loss in the {config}/{dtype} variant"
    mps = B * steps / dt
    bpm = ALGO_BYTES_PER_MESH[(config, dtype)]
    return {"workload": WORKLOADS[config] + f", {B} meshes/GPU, {dtype} storage", "value": mps, "unit": "meshes/s",
            "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup, "prewarm_steps": prewarm, "dtype": dtype,
            "final_loss": loss,
            "step_roofline": {"bound": "hbm", "achieved": mps * bpm / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": mps * bpm / 1e9 / HBM_PEAK_GBS, "note": f"meshes/s x {bpm / 1e6:.2f} MB/mesh (SURVEY 8(d))"}}


class RefBatch:
    """What cheb_VAE.forward reads of a torch_geometric Batch (cheb_VAE.py:195-200): .x [B*N, 3], .num_graphs, .edge_index."""

    def __init__(self, x):
        self.x, self.num_graphs, self.edge_index = x.reshape(-1, x.shape[-1]), x.shape[0], None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_loop_variant(dev, B, steps, warmup, prewarm, grad_mode="autograd"):
    """The reference's OWN call sequence around the drop-in modules, timed with the headline's protocol -- what an unchanged
    main.py gets (main.py:74-81 and :251): optimizer.zero_grad() -> model(data, x_gt, sex_hot, m_type="train") ->
    loss.backward() -> torch.optim.Adam(net.parameters(), lr, weight_decay=5e-4).step(); fp64 x_gt and int64 one-hot labels
    as main.py:69-71 hand them over; no engine.TrainStep, no fused optimizer."""
    net = build_model(dev).train()
    if grad_mode != "autograd":
        net.grad_mode = grad_mode            # the one-line opt-in of models/cheb_VAE.py (gradients assigned, not accumulated)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
    x = torch.randn(B, net.num_nodes[0], 3, generator=torch.Generator().manual_seed(0)).to(dev)
    x_gt = x.double()
    y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
    data = RefBatch(x)
    phase = [0.0] * 4

    def step(acc=False):
        t0 = time.perf_counter()
        opt.zero_grad()
        t1 = time.perf_counter()
        loss, correct, out, z, y_hat = net(data, x_gt, y, m_type="train")
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        opt.step()
        if acc:
            t4 = time.perf_counter()
            for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                phase[i] += v
        return loss
    for i in range(prewarm + warmup):
        step()
        if i % 50 == 49:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    loss = float(loss.detach())
    assert np.isfinite(loss), "non-finite loss in the reference-loop variant"
    n_ph = max(20, min(steps, 100))          # host time of each phase (the loop is host-bound), outside the timed region
    for _ in range(n_ph):
        step(True)
    torch.cuda.synchronize(dev)
    host_us = {k: round(1e6 * v / n_ph, 1) for k, v in zip(("zero_grad", "net()", "backward", "optimizer.step"), phase)}
    import meshvae_hip
    mps = B * steps / dt
    bpm = ALGO_BYTES_PER_MESH[("train5k", "f32")]
    return {"workload": WORKLOADS["train5k"] + f", {B} meshes/GPU, f32 -- driven by the reference's loop (main.py:74-81,251): "
                        "net(data, x_gt, y, m_type='train') -> loss.backward() -> torch.optim.Adam.step() -> zero_grad()",
            "value": mps, "unit": "meshes/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
            "prewarm_steps": prewarm, "dtype": "f32", "final_loss": loss, "optimizer": "torch.optim.Adam (torch default: foreach)",
            "async_launcher": meshvae_hip.launcher(dev.index) is not None, "grad_mode": grad_mode,
            "host_us": host_us, "host_cpu": cpu_model(),
            "note": "host-bound: the calling thread's time per step is the sum of host_us (it scales with the host CPU, not the GPU)",
            "step_roofline": {"bound": "hbm", "achieved": mps * bpm / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": mps * bpm / 1e9 / HBM_PEAK_GBS, "note": f"meshes/s x {bpm / 1e6:.2f} MB/mesh (SURVEY 8(d))"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="train5k")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the activations between layers (arithmetic accumulates in fp32 either way)")
    ap.add_argument("--batch", type=int, default=64, help="meshes per GPU")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (default: eager C++ launch sequence -- on ROCm 7.2 the "
                         "graph executor serialises the side-stream branch, eager overlaps it and is ~15%% faster)")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--micro", type=int, default=1, help="independent chains the per-GPU batch is pipelined over")
    ap.add_argument("--prewarm-steps", type=int, default=600,
                    help="untimed steps run BEFORE the --warmup steps (clock / power-state ramp of a cold GPU: the "
                         "first process on a fresh box was seen 25 %% slow otherwise); reported as prewarm_steps.  A step "
                         "COUNT, not a duration: every rank must issue the same number of gradient all-reduces.  0 disables")
    ap.add_argument("--rehearse-allreduce", action="store_true",
                    help="1 GPU: bring up a 1-rank RCCL group and run the gradient collective anyway (rehearsal of the "
                         "multi-GPU stream hand-over on a one-GPU box)")
    ap.add_argument("--ar-overlap", action="store_true", help="two-bucket overlapped gradient all-reduce (opt-in)")
    ap.add_argument("--seed", type=int, default=666, help="weights: this seed on every rank; noise: seed + rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra driver-timed legs of the default line (the reference's own loop; bf16 storage; the "
                         "20k configuration; inference latencies)")
    ap.add_argument("--only-reference-loop", action="store_true",
                    help="time ONLY the reference-API loop variant and print its object (A/B tooling, tools/ab_ref.sh)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (rank 0's JSON): libraries that print there (RCCL writes a five-line version
    # banner to stdout when the first communicator comes up) are pointed at stderr until that line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = args.rehearse_allreduce                             # 1-rank rehearsal of the RCCL path
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        # RCCL ("nccl") is the product backend.  MESHVAE_DIST_BACKEND=gloo is the REHEARSAL of this multi-rank branch on a
        # one-GPU box (tests/test_gpu_ddp.py: two ranks share cuda:0; RCCL refuses two ranks on one device)
        backend = os.environ.get("MESHVAE_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.config == "infer":
        assert world == 1, "--config infer is a single-GPU latency measurement"
        assert args.dtype == "f32", "--config infer runs the fp32 module path"
        return run_infer(args, dev, emit)

    if args.only_reference_loop:
        assert world == 1
        return emit(reference_loop_variant(dev, args.batch, args.steps, args.warmup, max(args.prewarm_steps, 0)))

    from meshvae_hip.engine import TrainStep
    net = build_model(dev, args.config)
    net.train()
    B = args.batch
    n0 = net.num_nodes[0]
    # weights: same seed on every rank AND a broadcast from rank 0 inside TrainStep; reparameterisation noise and
    # dropout masks: private generators seeded seed + rank, so no two ranks draw the same noise for their shards
    step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=bool(args.graph), m_type="train",
                     n_micro=args.micro, noise_seed=args.seed, rehearse_allreduce=args.rehearse_allreduce,
                     overlap_allreduce=args.ar_overlap, storage=args.dtype)
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, n0, 3, generator=g)
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

    prewarm = max(args.prewarm_steps, 0) if args.config == "train5k" else min(max(args.prewarm_steps, 0), 60)
    for i in range(prewarm):   # not part of the contract's W warm-up steps: extra untimed work, reported as prewarm_steps
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
        bytes_per_mesh = ALGO_BYTES_PER_MESH[(args.config, args.dtype)]
        precision = ("fp32 storage and arithmetic (the reference's dtype; --dtype bf16 is configs[1] as worded)"
                     if args.dtype == "f32" else
                     "activations and their gradients stored bf16 in HBM between layers, fp32 master weights and fp32 "
                     "accumulation everywhere (fp32 recurrence state in LDS)")
        out = {
            "metric": "meshes/sec fwd+bwd, 5k-vertex ChebConv VAE" if args.config == "train5k" else
                      "meshes/sec fwd+bwd, 20k-vertex K=10 ChebConv VAE",
            "value": meshes_per_s, "unit": "meshes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": prewarm,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config] + f", {B} meshes/GPU, {args.dtype} storage",
                       "global_batch": world * B, "per_gpu_batch": B, "vertices": n0,
                       "parallelism": f"dp{world}", "hipgraph": bool(step.use_graph),
                       "micro_batches": step.n_micro, "precision": precision},
            "step_roofline": {"bound": "hbm", "achieved": meshes_per_s / world * bytes_per_mesh / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": meshes_per_s / world * bytes_per_mesh / 1e9 / HBM_PEAK_GBS,
                              "note": f"whole step per GPU: meshes/s x {bytes_per_mesh / 1e6:.2f} MB/mesh (SURVEY 8(d))"},
            "final_loss": loss,
        }
        if not args.no_kernel_roofline:
            out["roofline"], out["kernels"] = kernel_roofline(net, B, dev, args.config, args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, B if args.config == "train5k" else 4)
        if world == 1 and args.config == "train5k" and args.dtype == "f32" and not args.no_variants and not step.use_graph:
            # the other BASELINE configurations this library runs, timed by the same process with the same protocol AFTER
            # the headline's timed region (they never touch `value`): configs[1] as worded (bf16 storage), configs[3]
            del step
            torch.cuda.empty_cache()
            out["variants"] = {
                "reference_loop": reference_loop_variant(dev, B, args.steps, args.warmup, min(prewarm, 100)),
                "bf16": timed_variant(dev, "train5k", "bf16", B, args.steps, args.warmup, min(prewarm, 100), args.seed),
                "hires20k": timed_variant(dev, "hires20k", "f32", B, max(10, args.steps // 2), max(3, args.warmup // 2), 20,
                                          args.seed)}
            # the same loop with the module's one-line opt-in net.grad_mode = "assign" (models/cheb_VAE.py: gradients
            # assigned to .grad, not accumulated through 29 AccumulateGrad nodes): reported beside, never instead
            ra = reference_loop_variant(dev, B, args.steps, args.warmup, min(prewarm, 100), grad_mode="assign")
            out["variants"]["reference_loop"]["grad_mode_assign"] = {k: ra[k] for k in ("value", "ms_per_step", "host_us", "final_loss")}
            # ... and configs[4] (inference, hipGraph replay): the latencies of `--config infer`
            _, lat, eager, it = infer_latencies(dev, max(20, args.steps), max(3, args.warmup // 2))
            _, lat2, _, _ = infer_latencies(dev, max(20, args.steps), max(3, args.warmup // 2), batched=True)
            out["variants"]["infer"] = {"workload": WORKLOADS["infer"], "unit": "ms", "steps": it,
                                        "latency_ms": {f"b{b}": lat[b] for b in lat},
                                        "eager_latency_ms": {f"b{b}": eager[b] for b in eager},
                                        "replay_equals_eager_bitwise": True,
                                        # (both decodes in one 2B-mesh pass, crecon_ops.estimate_diff_device: bitwise the two calls)
                                        "batched_decode_latency_ms": {f"b{b}": lat2[b] for b in lat2}}
        emit(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
