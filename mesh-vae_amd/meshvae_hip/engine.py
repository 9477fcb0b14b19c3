"""Train-step engine around the model: flat parameter/gradient buffers, data-parallel
gradient all-reduce (RCCL over xGMI via torch.distributed), fused Adam and hipGraph replay.

The reference has no distributed code at all; meshes are independent
through the whole forward/backward, so the batch is sharded over ranks and the ONLY exchange
is one sum all-reduce of the flat gradient buffer (712,642 fp32 = 2.85 MB at default.cfg)
per step, followed by the 1/world scale folded into the optimizer kernel.  Parameters and
gradients live in two contiguous buffers so no bucketing copies exist; `dec_lin_1` (never
used, cheb_VAE.py:165) stays in the buffers with a zero gradient.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import check, lib


def shard_range(global_batch, rank, world):
    """Contiguous, balanced [lo, hi) slice of the global batch owned by `rank`."""
    if global_batch < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard request")
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatParams:
    """Re-homes every parameter (and its .grad) of `module` as a view into one flat buffer.
    Every tensor starts on a 256-byte boundary (the kernels take 16-byte vector loads of weight
    rows only from aligned bases); the padding holds zeros in both buffers and stays zero under
    Adam, so `numel` (padded) is what the all-reduce and the optimizer run over and `n_params`
    is the model's parameter count."""

    ALIGN = 64  # floats

    def __init__(self, module):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        dev, dtype = params[0].device, params[0].dtype
        self.n_params = sum(p.numel() for p in params)
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.numel = off
        self.param = torch.zeros(self.numel, dtype=dtype, device=dev)
        self.grad = torch.zeros(self.numel, dtype=dtype, device=dev)
        self.names = [n for n, _ in module.named_parameters()]
        for p, off in zip(params, offs):
            n = p.numel()
            self.param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.param[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
        self.params, self.offsets = params, offs

    def zero_grad(self):
        self.grad.zero_()

    def conv_dense_split(self):
        """Offset (floats) where the dense-layer parameters start in the flat buffers: cheb_VAE registers its
        ChebConv lists first and the nn.Linear layers after them (cheb_VAE.py:121-166), so the dense gradients
        -- final half-way through the backward -- are one contiguous tail.  None if the order is different."""
        conv = [n.startswith(("cheb.", "cheb_dec.")) for n in self.names]
        k = conv.index(False) if False in conv else len(conv)
        if k == 0 or k == len(conv) or any(conv[k:]):
            return None
        return self.offsets[k]

    NO_GRAD = ("dec_lin_1.",)   # parameters the forward never touches (cheb_VAE.py:165): .grad stays None in the reference

    def no_grad_range(self):
        """[lo, hi) floats of the flat buffers covered by parameters that never receive a gradient (contiguous
        in cheb_VAE's registration order); (0, 0) when there are none."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(self.NO_GRAD)]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return (0, 0)
        hi = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.offsets) else self.numel
        return (self.offsets[idx[0]], hi)

    def all_reduce(self, group=None, always=False):
        """Sum all-reduce of the flat gradient buffer (one collective per step).  `always` runs the collective
        on a 1-rank group too: the only way to rehearse the RCCL stream hand-over of the multi-GPU path on a
        one-GPU box (bench.py --rehearse-allreduce)."""
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or always):
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            return dist.get_world_size(group)
        return 1

    def broadcast(self, group=None, src=0):
        """Every rank takes rank `src`'s parameters (one collective over the flat buffer).  Data-parallel
        replicas must start identical; seeding every rank alike gives that only as long as nothing else
        has consumed the generator, so the engine does not rely on it."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.param, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
            return True
        return False


def rank_generators(seed, rank, device=None):
    """Per-rank noise streams of a data-parallel job: (host generator for the reparameterisation noise, device
    generator for the dropout uniforms or None on a CPU-only caller), both seeded `seed + rank` so that no two
    ranks draw the same eps / masks for their shards (same seed everywhere = every rank would see the SAME
    noise on different meshes, i.e. correlated gradient noise across the global batch)."""
    host = torch.Generator().manual_seed(int(seed) + int(rank))
    dev_gen = None
    if device is not None and torch.device(device).type == "cuda":
        dev_gen = torch.Generator(device=device).manual_seed(int(seed) + int(rank))
    return host, dev_gen


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps, weight_decay) semantics (reference main.py:251) as one
    HIP kernel over the flat buffers (mvh_adam_step)."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, device_counter=False):
        """device_counter: always use the replayable form (step number on the device, advanced by a tick kernel);
        default: only inside a stream capture."""
        if not flat.param.is_cuda:
            raise RuntimeError("FusedAdam runs on MI355X only (there is no CPU fallback)")
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.device_counter = bool(device_counter)
        # [lo, hi) of the flat buffer that gets no update: parameters without a gradient (torch.optim.Adam skips
        # `p.grad is None`; cheb_VAE's unused dec_lin_1 stays at its initial values in the reference)
        self.skip = flat.no_grad_range()
        self.exp_avg = torch.zeros_like(flat.param)
        self.exp_avg_sq = torch.zeros_like(flat.param)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=flat.param.device)

    def step(self, grad_scale=1.0):
        f = self.flat
        dev = f.param.device
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            args = (f.param.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), f.numel,
                    self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, float(grad_scale),
                    self.step_count.data_ptr())
            if torch.cuda.is_current_stream_capturing() or self.device_counter:
                # replayable form: the step number lives on the device and a tick kernel advances it
                self._host_step = None
                check(lib().mvh_adam_step(st, *args, *self.skip))
            else:
                # eager: the host counts (one launch less); the kernel keeps the device counter in sync
                if getattr(self, "_host_step", None) is None:
                    self._host_step = int(self.step_count.item())     # once: after construction / after graph replays
                self._host_step += 1
                check(lib().mvh_adam_step_counted(st, *args, self._host_step, *self.skip))


def scheduled_lr(config, epoch, current_lr):
    """The step-wise learning-rate table of the reference's epoch loop (main.py:266-269): every pair
    (learning_rates_epochs[i], learning_rates[i]) with `epoch > learning_rates_epochs[i]` overwrites the
    rate, in list order (so the last matching pair wins); no match leaves `current_lr` unchanged."""
    lr = current_lr
    for i, e in enumerate(config.get("learning_rates_epochs", [])):
        if epoch > e:
            lr = config["learning_rates"][i]
    return lr


class _Batch:
    def __init__(self, x):
        self.x, self.num_graphs, self.edge_index = x.reshape(-1, x.shape[-1]), x.shape[0], None


class TrainStep:
    """One data-parallel train step: forward + backward (+ all-reduce) + Adam.

    Eager (`use_graph=False`, the default) is the fast path on ROCm 7.2: the C++ launch sequence keeps the host ahead
    of the GPU and may fork its weight-gradient lanes freely (0.57 vs 0.62 ms per step at B = 64, profiles/r02_h).
    With `use_graph=True` the forward/backward and the optimizer are captured into hipGraphs
    (static shapes: fixed N, fixed per-rank B) and replayed; the gradient all-reduce runs
    between the two graphs on the same stream.  Everything random is drawn OUTSIDE the graphs into static device
    buffers before each replay: the reparameterisation noise on the host generator, as the reference does
    (cheb_VAE.py:316), and the dropout uniforms on the device generator (a generator consumed inside a capture would
    either raise or bake one offset -- one mask -- into every replay).
    """

    U_AHEAD = 16     # eager: steps' worth of dropout uniforms drawn by one generator launch

    def __init__(self, net, batch, lr=1e-3, weight_decay=5e-4, use_graph=False, m_type="train", group=None,
                 native=True, n_micro=1, noise_seed=None, rehearse_allreduce=False, overlap_allreduce=False,
                 storage="f32"):
        """noise_seed: reparameterisation noise and dropout uniforms come from generators private to this step,
        seeded `noise_seed + rank` (rank_generators).  None on a single rank keeps the reference's behaviour --
        the process-wide default generators (cheb_VAE.py:316) -- and on a multi-rank group means 666.
        rehearse_allreduce: run the gradient collective even on a 1-rank group; overlap_allreduce: the
        two-bucket form of _all_reduce_overlapped (opt-in until an 8-GPU measurement says otherwise).
        storage: "f32" (the reference's dtype) or "bf16": activations and their gradients between the conv layers
        are stored bf16 in HBM, arithmetic accumulates in fp32 (BASELINE configs[1] as worded; native step only)."""
        if storage not in ("f32", "bf16"):
            raise ValueError("storage must be 'f32' or 'bf16'")
        if storage == "bf16" and not native:
            raise ValueError("bf16 storage exists in the native step only")
        self.storage = storage
        self.net, self.B, self.m_type, self.group = net, batch, m_type, group
        self.dev = next(net.parameters()).device
        self.flat = FlatParams(net)
        self.flat.broadcast(group)                  # replicas start from rank 0's parameters, whatever the seeds were
        self.opt = FusedAdam(self.flat, lr=lr, weight_decay=weight_decay)
        self.rehearse_allreduce, self.overlap_allreduce = bool(rehearse_allreduce), bool(overlap_allreduce)
        world_now = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if noise_seed is None and world_now > 1:
            noise_seed = 666
        self.host_gen = self.dev_gen = None
        if noise_seed is not None:
            rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
            self.host_gen, self.dev_gen = rank_generators(noise_seed, rank, self.dev)
        n0, f0 = net.num_nodes[0], net.filters[0]
        self.x = torch.zeros(batch, n0, f0, device=self.dev)
        self.x_gt = torch.zeros(batch, n0, f0, device=self.dev)
        # one-hot labels as fp32 (what the kernels read): copy_() from the loader's int64 labels converts once
        self.y = torch.zeros(batch, net.num_class, dtype=torch.float32, device=self.dev)
        self.eps = torch.zeros(batch, net.z, device=self.dev)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # Hardware-queue budget (profiles/r05_fourth_queue.txt): the step keeps three queues busy (caller's stream + two
        # weight-gradient lanes) -- fine under the runtime's default of 4; a process group adds the collective library's
        # streams, and with 4 queues the lanes then share the main chain's (0.59-0.61 instead of 0.47 ms per step with a
        # 1-rank RCCL group).  The variable is read when the HIP runtime starts, so it is the launching script's to set.
        if (self.world > 1 or rehearse_allreduce) and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
            import warnings
            warnings.warn("meshvae_hip.TrainStep with a process group: set GPU_MAX_HW_QUEUES=8 in the environment of every "
                          "rank before the first torch.cuda call (the default of 4 hardware queues makes the step's "
                          "weight-gradient lanes share the main chain's queue: ~30 % slower steps)", RuntimeWarning)
        self.use_graph = use_graph
        self.graph_fb = self.graph_opt = None
        self._out = None
        self._comm = None      # stream of the overlapped dense-gradient all-reduce (created on first use)
        # static buffer (graph-safe), only for the batch size it was built for: any other call of the module
        # (a different B through net(...)) draws fresh host noise as the reference does
        # ... and only while this step keeps that buffer filled: with the draw-ahead of _draw_eps a step's noise is a view
        # of a device block and `self.eps` is never written, so a module-level train-mode call draws its own host noise
        net._eps_provider = lambda B, Z, device: (self.eps if (B == self.B and Z == self.eps.shape[1] and
                                                               not self._eps_ahead()) else None)
        net._prepare()                                         # topology upload must precede any capture
        # native=True: the whole forward+backward is one C++ launch sequence (mvh_vae_forward/backward);
        # native=False: the per-module autograd path (what main.py drives through model.forward).
        # n_micro > 1 splits the per-GPU batch into independent chains on their own streams: meshes
        # are independent, so the latency-bound small-level kernels of one chain overlap the
        # CU-filling level-0 kernels of another; gradients of the chains are summed before Adam.
        self.n_micro = n_micro if (native and batch % max(n_micro, 1) == 0) else 1
        self.native, self.streams, self.extra_grads, self.pool = None, [], [], None
        if native:
            mb = batch // self.n_micro
            self.native = []
            for j in range(self.n_micro):
                grads = None
                if j > 0:
                    buf = torch.zeros_like(self.flat.grad)
                    self.extra_grads.append(buf)
                    grads = [buf[off:off + p.numel()].view_as(p) for p, off in zip(self.flat.params, self.flat.offsets)]
                # chain 0 runs on the caller's stream, chains >= 1 on their own.  Eager: every chain is
                # enqueued by its own host thread (launching ~90 kernels costs the host ~0.4 ms, so one
                # thread cannot keep several chains fed) and forks its weight-gradient kernels to that
                # thread's internal side stream.  MEASURED (tools/host_time.py, tools/thread_probe.py):
                # the ROCm 7.2 runtime tops out at ~340 k launches/s over all host threads (225 k/s on
                # one), so two chains cost ~0.75 ms of launching and n_micro = 2 is slower (0.97 ms)
                # than one chain (0.90 ms); n_micro stays 1 by default.  hipGraph capture: single thread, chain 0 forks to a
                # torch side stream, chains >= 1 keep dW inline on their own stream, i.e. no captured stream that joined
                # the capture through an event ever forks a further stream.  Evidence (tools/graph_diag.py, one run per
                # scenario, raw HIP API with return codes + faulthandler): on ROCm 7.2 hipStreamEndCapture itself dies
                # with SIGSEGV -- before any hipGraphInstantiate -- for a nested fork made of three torch streams and
                # plain torch add kernels (main -> s1 -> s2, every branch joined; no code of this library involved),
                # and identically for the two-chain train step; single-level forks, also with a memset node on the
                # forked branch, capture (3 nodes / 2 edges), instantiate and replay with rc 0.  So the limitation is
                # the runtime's handling of second-level forks, not an unjoined branch or event reuse here; the
                # supported topology is covered by tests/test_gpu_engine.py::test_trainstep_graph_two_chains.
                chain = torch.cuda.Stream(self.dev) if j > 0 else None
                if use_graph:
                    side = chain if j > 0 else torch.cuda.Stream(self.dev)
                else:
                    side = None
                self.streams.append((chain, side))
                self.native.append(NativeStep(net, mb, grads=grads, side_stream=side, storage=storage))
            self._u_static = [torch.zeros((batch // self.n_micro) * nat.u_cols, device=self.dev) for nat in self.native]
            self._u_bufs, self._u_left, self._u_block = list(self._u_static), 0, None
            if self.n_micro > 1 and not use_graph:
                from concurrent.futures import ThreadPoolExecutor
                self.pool = ThreadPoolExecutor(max_workers=self.n_micro - 1, thread_name_prefix="meshvae-chain")

    def load(self, x, x_gt, y):
        self.x.copy_(x, non_blocking=True)
        self.x_gt.copy_(x_gt, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def _run_chain(self, j, u, cur):
        nat, st = self.native[j], self.streams[j][0]
        mb = self.B // self.n_micro
        sl = slice(j * mb, (j + 1) * mb)
        src = self._eps_view if getattr(self, "_eps_view", None) is not None else self.eps
        eps = src[sl] if self.m_type == "train" else None
        with torch.cuda.device(self.dev), torch.cuda.stream(st if st is not None else cur):
            return nat.forward_backward(self.x[sl], self.x_gt[sl], self.y[sl], eps, u)

    def _fwd_bwd(self):
        if self.native is not None:
            train = self.net.training and self.net.dropout.p > 0.0
            cur = torch.cuda.current_stream(self.dev)
            mb = self.B // self.n_micro
            # dropout uniforms of every chain: static buffers filled by _draw_noise() in chain order BEFORE this
            # function runs (never inside a hipGraph capture)
            us = [self._u_bufs[j] if train else None for j in range(self.n_micro)]
            for st, _ in self.streams:      # fork every chain BEFORE chain 0 queues its work on `cur`
                if st is not None:
                    st.wait_stream(cur)
            if self.pool is not None:
                futs = [self.pool.submit(self._run_chain, j, us[j], cur) for j in range(1, self.n_micro)]
                outs = [self._run_chain(0, us[0], cur)] + [f.result() for f in futs]
            else:
                outs = [self._run_chain(j, us[j], cur) for j in range(self.n_micro)]
            for j in range(1, self.n_micro):
                cur.wait_stream(self.streams[j][0])
                self.flat.grad.add_(self.extra_grads[j - 1])
            self._micro_outs = outs
            self._out = (outs[0][0], outs[0][1], outs[0][2]) if self.n_micro == 1 else None
            return
        self.flat.zero_grad()
        loss, correct, recon, extra, y_hat = self.net(_Batch(self.x), self.x_gt, self.y, m_type=self.m_type)
        loss.backward()
        self._out = (loss.detach(), correct, recon.detach())

    @property
    def out(self):
        """(loss, correct, recon) of the last step; micro-batched chains are assembled on first access."""
        if self._out is None and getattr(self, "_micro_outs", None):
            o = self._micro_outs
            self._out = (torch.stack([t[0] for t in o]).mean(), torch.stack([t[1] for t in o]).sum(),
                         torch.cat([t[2] for t in o], 0))
        return self._out

    def capture(self, warmup=3):
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):         # first call allocates workspaces / sets kernel attributes
                self._draw_noise()
                self._fwd_bwd()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            self._fwd_bwd()
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt):
            self.opt.step(1.0 / (self.world * self.n_micro))
        torch.cuda.synchronize(self.dev)

    def _draw_noise(self):
        """Everything random of one step, outside any graph: host noise (eps) and the dropout uniforms of every chain
        (device generator: the step's private one, or the process default when noise_seed is None)."""
        self._draw_eps()
        if self.native is not None and self.net.training and self.net.dropout.p > 0.0:
            # ONE generator launch per U_AHEAD steps (a 4 us kernel at the step boundary otherwise): a step takes its
            # uniforms from that block, chain by chain.  Eager: as views (no launch at all on 15 of 16 steps); hipGraph:
            # copied into the static buffers the captured launches read -- the same numbers in both modes
            pad = lambda k: -(-k // 64) * 64                 # every chain's piece starts on a 256-byte boundary
            n = sum(pad(b.numel()) for b in self._u_static)
            if self._u_left == 0:
                self._u_block = torch.rand(self.U_AHEAD * n, device=self.dev, generator=self.dev_gen)
                self._u_left = self.U_AHEAD
            off = (self.U_AHEAD - self._u_left) * n
            self._u_left -= 1
            bufs = []
            for b in self._u_static:
                view = self._u_block[off:off + b.numel()]
                off += pad(b.numel())
                if self.use_graph:
                    b.copy_(view)
                    bufs.append(b)
                else:
                    bufs.append(view)
            self._u_bufs = bufs

    E_AHEAD = 16     # eager, private host generator: steps' worth of reparameterisation noise per pinned copy

    def _eps_ahead(self):
        """Draw-ahead is only the same noise when nobody else draws from the generator in between: the step's PRIVATE host
        generator (noise_seed given), never the process default one (the reference's behaviour, cheb_VAE.py:316); and only
        where eps is passed per call (eager native chains; a hipGraph reads the static buffer)."""
        return (self.host_gen is not None and self.native is not None and not self.use_graph and
                os.environ.get("MESHVAE_EPS_AHEAD", "1") != "0")

    def _draw_eps(self):
        """Reparameterisation noise from the HOST default generator (reference cheb_VAE.py:316), moved
        with an asynchronous copy from a small ring of pinned buffers so the host never blocks on
        the device (a pageable .to(device) would serialise host and GPU every step)."""
        if self.m_type != "train":
            return
        if self._eps_ahead():
            # Private host generator, eager native step: the noise of E_AHEAD steps is drawn in one go -- the SAME
            # torch.normal calls in the same order, so every step sees the numbers it would have drawn itself -- and moved
            # by ONE pinned copy per E_AHEAD steps; a step's eps is a view of the device block.  (The per-step form costs a
            # host-to-device copy and an event record -- both system-scope operations -- on the critical chain of every step.)
            E = self.E_AHEAD
            if not hasattr(self, "_eps_blocks"):
                self._eps_blocks = [(torch.empty(E, self.B, self.net.z).pin_memory(), torch.cuda.Event(),
                                     torch.empty(E, self.B, self.net.z, device=self.dev)) for _ in range(2)]
                self._eps_blk_i, self._eps_left, self._eps_block = 0, 0, None
            if self._eps_left == 0:
                host, ev, devb = self._eps_blocks[self._eps_blk_i]
                self._eps_blk_i ^= 1
                ev.synchronize()                           # the copy that read this pinned block 2 E_AHEAD steps ago
                for i in range(E):
                    torch.normal(mean=0, std=1, size=(self.B, self.net.z), out=host[i], generator=self.host_gen)
                devb.copy_(host, non_blocking=True)        # (stream-ordered behind every step that read this device block)
                ev.record(torch.cuda.current_stream(self.dev))
                self._eps_block, self._eps_left = devb, E
            self._eps_view = self._eps_block[E - self._eps_left]
            self._eps_left -= 1
            return
        self._eps_view = None
        if not hasattr(self, "_eps_ring"):
            self._eps_ring = [(torch.empty(self.B, self.net.z).pin_memory(), torch.cuda.Event()) for _ in range(8)]
            self._eps_i = 0
        buf, ev = self._eps_ring[self._eps_i]
        self._eps_i = (self._eps_i + 1) % len(self._eps_ring)
        ev.synchronize()                                   # the copy that used this buffer 8 steps ago
        torch.normal(mean=0, std=1, size=(self.B, self.net.z), out=buf, generator=self.host_gen)
        self.eps.copy_(buf, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.dev))

    # MEASURED, not kept (round 2): drawing step t+1's noise (pinned copy + dropout uniforms) one step ahead on a stream
    # of its own, double-buffered, so that the step boundary on the main stream is Adam -> weight pack -> first
    # convolution: 598 vs 590 us per step in three alternating runs -- the fourth stream costs more (hardware-queue
    # sharing) than the ~15 us of copy / generator kernels it takes off the boundary.

    def set_epoch(self, config, epoch):
        """Apply the reference's LR table for `epoch` (main.py:266-269).  The rate is a kernel argument of the
        fused Adam launch, so it takes effect on the next eager step; a captured optimizer graph is re-captured."""
        lr = scheduled_lr(config, epoch, self.opt.lr)
        if lr != self.opt.lr:
            self.opt.lr = lr
            if self.use_graph and self.graph_opt is not None:
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt):
                    self.opt.step(1.0 / (self.world * self.n_micro))
        return lr

    def step(self):
        self._draw_noise()
        scale = 1.0 / (self.world * self.n_micro)   # every chain averaged over its own micro-batch
        if self.use_graph:
            if self.graph_fb is None:
                self.capture()
            self.graph_fb.replay()
            if self.n_micro > 1:
                self._out = None                # assembled from the chains' (static) outputs on the next access
            self.flat.all_reduce(self.group, always=self.rehearse_allreduce)
            self.graph_opt.replay()
            self.opt._host_step = None          # the replay advanced the device-side step counter
        else:
            self._fwd_bwd()
            if not self._all_reduce_overlapped():
                self.flat.all_reduce(self.group, always=self.rehearse_allreduce)
            self.opt.step(scale)
        return self.out

    def _all_reduce_overlapped(self):
        """Two-bucket gradient all-reduce of the eager native step: the dense-layer gradients (98 % of the bytes,
        one contiguous tail of the flat buffer) are final before the encoder half of the backward runs, so their
        collective is issued on its own stream behind the event the library records at that point
        (mvh_vae_wait_dense_grads) and runs underneath the rest of the backward; only the convolution weights
        (80 KB at default.cfg) are reduced after it.  The host has enqueued the whole backward long before the GPU
        gets there, so issuing the collective "late" from the host still starts it early on the device.  Same
        call order on every rank (dense bucket, then conv bucket).
        OPT-IN (overlap_allreduce=True; bench.py --ar-overlap).  Measured on one GPU with a 1-rank RCCL group (tools/dist_overhead3.sh):
        the second collective costs 17 us of fixed overhead per step (0.664 vs 0.647 ms), so it pays only where
        the 2.8 MB all-reduce takes clearly longer than an 80 KB one plus that -- which this one-GPU pool cannot
        measure; the default stays ONE collective after the backward."""
        if not (dist.is_available() and dist.is_initialized()) or self.native is None or self.n_micro != 1:
            return False
        if dist.get_world_size(self.group) <= 1 and not self.rehearse_allreduce:
            return False
        if not self.overlap_allreduce:
            return False
        split = self.flat.conv_dense_split()
        if split is None:
            return False
        cur = torch.cuda.current_stream(self.dev)
        if self._comm is None:
            self._comm = torch.cuda.Stream(self.dev)
        with torch.cuda.device(self.dev):
            check(lib().mvh_vae_wait_dense_grads(self._comm.cuda_stream))
            with torch.cuda.stream(self._comm):
                dist.all_reduce(self.flat.grad[split:], op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(self.flat.grad[:split], op=dist.ReduceOp.SUM, group=self.group)
            cur.wait_stream(self._comm)
        return True


class ClassifierStep:
    """One training step of the contrastive-reconstruction classifier (reference crecon.py:65-100):
    estimate_diff of the frozen VAE -> cheb_GCN -> cross-entropy -> backward -> Adam (coupled L2,
    crecon.py:311).  With `use_graph=True` the whole step is two hipGraphs over static buffers
    (diff/forward/backward, then the fused Adam); eager it is the same launch sequence driven from
    Python, which at these sizes is host-bound (tools/crecon_step.py).
    """

    def __init__(self, net, vae, batch, lr=1e-4, weight_decay=5e-4, use_graph=True, dtype="train"):
        from crecon_ops import estimate_diff_device
        self._diff = estimate_diff_device
        self.net, self.vae, self.B, self.dtype = net, vae, batch, dtype
        self.dev = next(net.parameters()).device
        self.flat = FlatParams(net)
        self.opt = FusedAdam(self.flat, lr=lr, weight_decay=weight_decay)
        n0 = vae.num_nodes[0]
        self.x = torch.zeros(batch, n0, 3, device=self.dev)
        self.label = torch.zeros(batch, dtype=torch.int64, device=self.dev)
        self.use_graph, self.graph_fb, self.graph_opt, self.out = use_graph, None, None, None
        vae._prepare()

    def load(self, x_gt, label):
        self.x.copy_(x_gt, non_blocking=True)          # .float() of the loader's fp64 x_gt (crecon.py:72)
        self.label.copy_(label, non_blocking=True)

    def _fwd_bwd(self):
        diff, vae_correct = self._diff(self.vae, self.x, self.label, self.dtype)
        self.flat.zero_grad()
        pred = self.net(diff)
        loss = torch.nn.functional.cross_entropy(pred, self.label)
        loss.backward()
        self.out = (loss.detach(), pred.detach(), vae_correct)

    def capture(self, warmup=3):
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._fwd_bwd()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            self._fwd_bwd()
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt):
            self.opt.step(1.0)
        torch.cuda.synchronize(self.dev)

    def step(self):
        """-> (loss, logits [B, num_classes], #meshes the VAE's own head classified correctly), device tensors."""
        if self.use_graph:
            if self.graph_fb is None:
                self.capture()
            self.graph_fb.replay()
            self.graph_opt.replay()
            self.opt._host_step = None          # the replay advanced the device-side step counter
        else:
            self._fwd_bwd()
            self.opt.step(1.0)
        return self.out


class NativeStep:
    """cheb_VAE.forward + loss.backward() through mvh_vae_forward / mvh_vae_backward: every
    kernel is enqueued from C++, activations live in one workspace, parameter gradients are
    written straight into the module's (flat) .grad views, weight-gradient kernels run on a
    side stream.  Semantics are those of `cheb_VAE.forward(data, x_gt, y, m_type)`."""

    def __init__(self, net, batch, grads=None, side_stream=None, storage="f32"):
        """grads: optional list of gradient tensors (one per parameter, default = the .grad views);
        side_stream: torch stream for the weight-gradient kernels (default: one internal per device);
        storage: "f32" | "bf16" storage of the activations between the conv layers (mvh_vae_desc_t.storage)."""
        import ctypes
        from . import VaeDesc
        self.side = side_stream
        from .functional import _need_gpu  # noqa: F401
        net._prepare()
        self.net, self.B = net, batch
        self.dev = next(net.parameters()).device
        n = net.n_layers
        d = VaeDesc()
        d.n_layers, d.num_features = n, net.filters[0]
        d.num_hidden, d.num_classes, d.num_style = net.num_hidden, net.num_class, net.z
        d.dropout_p = float(net.dropout.p)
        from . import STORAGE_BF16, STORAGE_F32
        d.storage = STORAGE_BF16 if storage == "bf16" else STORAGE_F32
        for i, v in enumerate(net.filters):
            d.filters[i] = v
        for i in range(n + 1):
            d.K[i] = net.K[i]
            d.num_nodes[i] = net.num_nodes[i]
        for i in range(n):
            d.lap[i], d.lap_t[i] = net._lap[i].fwd.struct, net._lap[i].bwd.struct
            d.down[i], d.down_t[i] = net._down[i].fwd.struct, net._down[i].bwd.struct
            d.up[i], d.up_t[i] = net._up[i].fwd.struct, net._up[i].bwd.struct
        d.lap[n], d.lap_t[n] = net._lap_final.fwd.struct, net._lap_final.bwd.struct
        # vertex-patch plans (csrc/cheb_patch.hip): the 16 -> 16 decoder stage of a 2 049 .. 5 119-vertex level runs as
        # (mesh, vertex patch) workgroups; the plan hangs off the desc's COPY of the level's Laplacian (fp32 storage)
        self._patch_keep = []
        from . import topology
        for i in range(n):
            j = n - 1 - i                                    # decoder stage of level i: filters[i + 2] -> filters[i + 1]
            if net.filters[i + 1] == 16 and net.filters[i + 2] == 16:       # (bf16 storage: the backward only)
                got = topology.patch_plan(net._lap[i], int(net.K[j]) - 1, net._up[i])
                if got is not None:
                    self._patch_keep.append(got)
                    d.lap[i].patch = d.lap_t[i].patch = ctypes.addressof(got[0])
        # ... and the first layer (<= 4 -> 16 channels in front of its one-hot downsampling): the plan whose pooling rows are
        # that operator's hangs off the desc's copy of the operator (k_patch_enc0; either storage)
        if n >= 1 and net.filters[0] <= 4 and net.filters[1] == 16:
            got = topology.patch_plan(net._lap[0], int(net.K[0]) - 1, down_op=net._down[0])
            if got is not None and got[0].n_pool_rows == net._down[0].fwd.n_rows:
                self._patch_keep.append(got)
                d.down[0].patch = ctypes.addressof(got[0])
        self.desc = d
        L = lib()
        self.params = [p for _, p in net.named_parameters()]
        assert len(self.params) == L.mvh_vae_param_count(ctypes.byref(d)), "unexpected parameter list"
        if grads is None:                    # default: gradients land in the parameters' own .grad tensors
            for p in self.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        PtrArr = ctypes.c_void_p * len(self.params)
        self._P = PtrArr(*[p.data_ptr() for p in self.params])
        if isinstance(grads, str) and grads == "external":
            # the caller hands a gradient table to run_backward per call (the module path's autograd node): this object
            # owns none, and the entry points that would use one refuse
            self.grads, self._G = None, None
        else:
            self.grads = grads if grads is not None else [p.grad for p in self.params]
            self._G = PtrArr(*[g.data_ptr() for g in self.grads])
        self.ws_bytes = L.mvh_vae_step_ws_bytes(ctypes.byref(d), batch)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
        B, C, Z = batch, net.num_class, net.z
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.recon = torch.empty(B, net.num_nodes[0], net.filters[0], **f32)
        self.kld, self.z_ = torch.empty(B, **f32), torch.empty(B, Z, **f32)
        self.y_hat, self.mu, self.logvar = torch.empty(B, C, **f32), torch.empty(B, Z, **f32), torch.empty(B, Z, **f32)
        self.correct = torch.empty((), dtype=torch.int64, device=self.dev)
        self.y_f = torch.empty(B, C, **f32)
        self.u_cols = 3 * net.num_hidden + net.dec_lin_2.out_features
        self._loss = {}
        from models.cheb_VAE import LOG_SIGMA
        self.log_sigma = LOG_SIGMA

    def backward(self, x, x_gt, y_f, d_loss, eps=None, drop_u=None):
        """loss.backward() for the forward this object ran last (activations live in its workspace): the gradient
        of every parameter times the scalar `d_loss` (a 0-d device tensor of the loss dtype) goes to `self.grads`."""
        L = lib()
        if self._G is None:
            raise RuntimeError("this NativeStep has no gradient table of its own (grads='external'): use run_backward")
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        f64 = int(x_gt.dtype == torch.float64)
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            check(L.mvh_vae_backward(st, ctypes.byref(self.desc), self._P, self._G, x.data_ptr(), y_f.data_ptr(),
                                     x_gt.data_ptr(), f64, ptr(eps), ptr(drop_u), self.B, self.log_sigma,
                                     d_loss.data_ptr(), self.recon.data_ptr(), self.y_hat.data_ptr(), self.mu.data_ptr(),
                                     self.logvar.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                     self.side.cuda_stream if self.side is not None else None))

    def encode(self, x, drop_u_enc=None):
        """net.encoder(x) (cheb_VAE.py:261-273) as one native launch sequence -> h [B, num_hidden] (new tensor)."""
        h = torch.empty(self.B, self.net.num_hidden, dtype=torch.float32, device=self.dev)
        x = x.contiguous()
        with torch.cuda.device(self.dev):
            check(lib().mvh_vae_encode(torch.cuda.current_stream(self.dev).cuda_stream, ctypes.byref(self.desc), self._P,
                                       x.data_ptr(), None if drop_u_enc is None else drop_u_enc.data_ptr(), self.B,
                                       h.data_ptr(), self.ws.data_ptr(), self.ws_bytes))
        return h

    def decode(self, zy, drop_u=None):
        """net.decoder(zy) (cheb_VAE.py:275-292) as one native launch sequence -> recon [B, N, F] (new tensor)."""
        recon = torch.empty(self.B, self.net.num_nodes[0], self.net.filters[0], dtype=torch.float32, device=self.dev)
        zy = zy.contiguous().to(torch.float32)
        with torch.cuda.device(self.dev):
            check(lib().mvh_vae_decode(torch.cuda.current_stream(self.dev).cuda_stream, ctypes.byref(self.desc), self._P,
                                       zy.data_ptr(), None if drop_u is None else drop_u.data_ptr(), self.B,
                                       recon.data_ptr(), self.ws.data_ptr(), self.ws_bytes))
        return recon

    def ws_tensor(self, name, index=0):
        """Test aid: a flat fp32 view of one activation / gradient tensor of the last step inside the workspace
        (mvh_vae_ws_offset; fp32 storage only)."""
        cnt = ctypes.c_int64(0)
        off = lib().mvh_vae_ws_offset(ctypes.byref(self.desc), self.B, name.encode(), index, ctypes.byref(cnt))
        if off < 0:
            raise KeyError((name, index))
        return self.ws[off:off + 4 * cnt.value].view(torch.float32)

    def _refresh_pointers(self):
        for i, p in enumerate(self.params):
            self._P[i] = p.data_ptr()
            if self._G is not None:
                self._G[i] = self.grads[i].data_ptr()

    def refresh_param_pointers(self):
        """The parameter table only (the module path hands its own gradient table to run_backward): one list of
        data_ptr() calls, written to the ctypes array only when a parameter moved (load_state_dict copies in place;
        .to(), FlatParams or an assignment to p.data re-home it)."""
        ptrs = [p.data_ptr() for p in self.params]
        if ptrs != getattr(self, "_P_seen", None):
            self._P[:] = ptrs
            self._P_seen = ptrs

    def run_forward(self, x, x_gt, y_f, eps, drop_u, outs, launcher=None):
        """mvh_vae_forward into caller-provided output tensors `outs` = (loss, correct, recon, kld, rec, z_, y_hat, mu,
        logvar) -- loss / rec of x_gt's dtype.  launcher: the device's asynchronous launcher (meshvae_hip.launcher) or
        None for the synchronous call on the current stream; stream semantics are the same either way."""
        L = lib()
        loss, correct, recon, kld, rec, z_, y_hat, mu, logvar = outs
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        args = (ctypes.byref(self.desc), self._P, x.data_ptr(), y_f.data_ptr(), x_gt.data_ptr(),
                int(x_gt.dtype == torch.float64), ptr(eps), ptr(drop_u), self.B, self.log_sigma, loss.data_ptr(),
                correct.data_ptr(), recon.data_ptr(), kld.data_ptr(), rec.data_ptr(), z_.data_ptr(), y_hat.data_ptr(),
                mu.data_ptr(), logvar.data_ptr(), self.ws.data_ptr(), self.ws_bytes)
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            if launcher is not None:
                check(L.mvh_vae_forward_async(launcher, st, *args))
            else:
                check(L.mvh_vae_forward(st, *args))

    def run_backward(self, x, x_gt, y_f, eps, drop_u, d_loss, recon, y_hat, mu, logvar, G, launcher=None):
        """mvh_vae_backward for the forward this object ran last, gradients to the pointer table `G` (a ctypes array
        of one address per parameter, state_dict order)."""
        L = lib()
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        args = (ctypes.byref(self.desc), self._P, G, x.data_ptr(), y_f.data_ptr(), x_gt.data_ptr(),
                int(x_gt.dtype == torch.float64), ptr(eps), ptr(drop_u), self.B, self.log_sigma, ptr(d_loss),
                recon.data_ptr(), y_hat.data_ptr(), mu.data_ptr(), logvar.data_ptr(), self.ws.data_ptr(), self.ws_bytes)
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            if launcher is not None:
                check(L.mvh_vae_backward_async(launcher, st, *args))
            else:
                check(L.mvh_vae_backward(st, *args, self.side.cuda_stream if self.side is not None else None))

    def _outs(self, dtype):
        if dtype not in self._loss:
            self._loss[dtype] = (torch.empty((), dtype=dtype, device=self.dev),
                                 torch.empty(self.B, dtype=dtype, device=self.dev))
        return self._loss[dtype]

    def forward_backward(self, x, x_gt, y, eps=None, drop_u=None, backward=True):
        """x [B,N,F] fp32, x_gt fp32/fp64, y int64 or float one-hot [B,C].  Returns
        (loss, correct, recon, [kld, rec, z_], y_hat) exactly like cheb_VAE.forward."""
        import ctypes
        L = lib()
        if backward and self._G is None:
            raise RuntimeError("this NativeStep has no gradient table of its own (grads='external'): use run_backward")
        x, x_gt = x.contiguous(), x_gt.contiguous()
        if y.dtype == torch.float32 and y.is_contiguous() and y.device == self.dev:
            y_f = y                      # already what the kernels read: no per-step conversion launch
        else:
            self.y_f.copy_(y)
            y_f = self.y_f
        loss, rec = self._outs(x_gt.dtype)
        f64 = int(x_gt.dtype == torch.float64)
        d = ctypes.byref(self.desc)
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            if backward:     # the input-only part of the backward runs underneath the forward
                check(L.mvh_vae_backward_prefetch(st, d, x.data_ptr(), self.B, self.ws.data_ptr(), self.ws_bytes,
                                                  self.side.cuda_stream if self.side is not None else None))
            check(L.mvh_vae_forward(st, d, self._P, x.data_ptr(), y_f.data_ptr(), x_gt.data_ptr(), f64, ptr(eps),
                                    ptr(drop_u), self.B, self.log_sigma, loss.data_ptr(), self.correct.data_ptr(),
                                    self.recon.data_ptr(), self.kld.data_ptr(), rec.data_ptr(), self.z_.data_ptr(),
                                    self.y_hat.data_ptr(), self.mu.data_ptr(), self.logvar.data_ptr(),
                                    self.ws.data_ptr(), self.ws_bytes))
            if backward:
                check(L.mvh_vae_backward(st, d, self._P, self._G, x.data_ptr(), y_f.data_ptr(), x_gt.data_ptr(),
                                         f64, ptr(eps), ptr(drop_u), self.B, self.log_sigma, None, self.recon.data_ptr(),
                                         self.y_hat.data_ptr(), self.mu.data_ptr(), self.logvar.data_ptr(),
                                         self.ws.data_ptr(), self.ws_bytes,
                                         self.side.cuda_stream if self.side is not None else None))
        return loss, self.correct, self.recon, [self.kld, rec, self.z_], self.y_hat
