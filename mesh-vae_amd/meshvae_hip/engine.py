"""Train-step engine around the model: flat parameter/gradient buffers, data-parallel
gradient all-reduce (RCCL over xGMI via torch.distributed), fused Adam and hipGraph replay.

The reference has no distributed code at all; meshes are independent
through the whole forward/backward, so the batch is sharded over ranks and the ONLY exchange
is one sum all-reduce of the flat gradient buffer (712,642 fp32 = 2.85 MB at default.cfg)
per step, followed by the 1/world scale folded into the optimizer kernel.  Parameters and
gradients live in two contiguous buffers so no bucketing copies exist; `dec_lin_1` (never
used, cheb_VAE.py:165) stays in the buffers with a zero gradient.
"""
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
    """Re-homes every parameter (and its .grad) of `module` as a view into one flat buffer."""

    def __init__(self, module):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        dev, dtype = params[0].device, params[0].dtype
        self.numel = sum(p.numel() for p in params)
        self.param = torch.empty(self.numel, dtype=dtype, device=dev)
        self.grad = torch.zeros(self.numel, dtype=dtype, device=dev)
        self.names = [n for n, _ in module.named_parameters()]
        off = 0
        for p in params:
            n = p.numel()
            self.param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.param[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
            off += n
        self.params = params

    def zero_grad(self):
        self.grad.zero_()

    def all_reduce(self, group=None):
        """Sum all-reduce of the flat gradient buffer (one collective per step)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            return dist.get_world_size(group)
        return 1


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps, weight_decay) semantics (reference main.py:251) as one
    HIP kernel over the flat buffers (mvh_adam_step)."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not flat.param.is_cuda:
            raise RuntimeError("FusedAdam runs on MI355X only (there is no CPU fallback)")
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.param)
        self.exp_avg_sq = torch.zeros_like(flat.param)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=flat.param.device)

    def step(self, grad_scale=1.0):
        f = self.flat
        with torch.cuda.device(f.param.device):
            check(lib().mvh_adam_step(torch.cuda.current_stream(f.param.device).cuda_stream, f.param.data_ptr(),
                                      f.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                      f.numel, self.lr, self.betas[0], self.betas[1], self.eps,
                                      self.weight_decay, float(grad_scale), self.step_count.data_ptr()))


class _Batch:
    def __init__(self, x):
        self.x, self.num_graphs, self.edge_index = x.reshape(-1, x.shape[-1]), x.shape[0], None


class TrainStep:
    """One data-parallel train step: forward + backward (+ all-reduce) + Adam.

    With `use_graph=True` the forward/backward and the optimizer are captured into hipGraphs
    (static shapes: fixed N, fixed per-rank B) and replayed; the gradient all-reduce runs
    between the two graphs on the same stream.  The reparameterisation noise is still drawn on
    the host default generator every step, as the reference does (cheb_VAE.py:316), and copied
    into a static device buffer before replay.
    """

    def __init__(self, net, batch, lr=1e-3, weight_decay=5e-4, use_graph=True, m_type="train", group=None):
        self.net, self.B, self.m_type, self.group = net, batch, m_type, group
        self.dev = next(net.parameters()).device
        self.flat = FlatParams(net)
        self.opt = FusedAdam(self.flat, lr=lr, weight_decay=weight_decay)
        n0, f0 = net.num_nodes[0], net.filters[0]
        self.x = torch.zeros(batch, n0, f0, device=self.dev)
        self.x_gt = torch.zeros(batch, n0, f0, device=self.dev)
        self.y = torch.zeros(batch, net.num_class, dtype=torch.int64, device=self.dev)
        self.eps = torch.zeros(batch, net.z, device=self.dev)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.use_graph = use_graph
        self.graph_fb = self.graph_opt = None
        self.out = None
        net._eps_provider = lambda B, Z, device: self.eps      # static buffer (graph-safe)
        net._prepare()                                         # topology upload must precede any capture

    def load(self, x, x_gt, y):
        self.x.copy_(x, non_blocking=True)
        self.x_gt.copy_(x_gt, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def _fwd_bwd(self):
        self.flat.zero_grad()
        loss, correct, recon, extra, y_hat = self.net(_Batch(self.x), self.x_gt, self.y, m_type=self.m_type)
        loss.backward()
        self.out = (loss.detach(), correct, recon.detach())

    def capture(self, warmup=3):
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):         # first call allocates workspaces / sets kernel attributes
                self._draw_eps()
                self._fwd_bwd()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            self._fwd_bwd()
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt):
            self.opt.step(1.0 / self.world)
        torch.cuda.synchronize(self.dev)

    def _draw_eps(self):
        if self.m_type == "train":
            host = torch.normal(mean=0, std=1, size=(self.B, self.net.z))     # host RNG (reference :316)
            self.eps.copy_(host, non_blocking=False)

    def step(self):
        self._draw_eps()
        if self.use_graph:
            if self.graph_fb is None:
                self.capture()
            self.graph_fb.replay()
            self.flat.all_reduce(self.group)
            self.graph_opt.replay()
        else:
            self._fwd_bwd()
            self.flat.all_reduce(self.group)
            self.opt.step(1.0 / self.world)
        return self.out
