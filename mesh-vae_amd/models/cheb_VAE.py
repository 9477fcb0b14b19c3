"""`models.cheb_VAE` -- MI355X-native conditional mesh VAE with the reference's API.

Same constructor, attributes, method names, return tuples and state_dict layout as the
reference class (models/cheb_VAE.py:104-351) so main.py / inference.py / crecon.py run
unchanged, but every stage executes hand-written HIP kernels from libmeshvae_hip:

  encoder   4 x [ChebConv+ReLU (fused) -> one-hot downsample gather] -> enc_lin+ReLU+dropout
  head      classifier softmax, z_mean / z_log_var, reparameterisation   (one fused kernel)
  decoder   dec_lin, dec_lin_2 (+ReLU+dropout) -> 4 x [barycentric upsample -> ChebConv+ReLU]
            -> final ChebConv on the coarsest edge list (the reference's quirk, :288)
  loss      KLD + Gaussian NLL(fixed sigma) - 2 log q(y)                   (fused kernels)

Topology (A/D/U) is converted once to device CSR at construction.  Parameters are created
in the reference's order, so `torch.manual_seed(s)` yields bit-identical initial weights.
The unused pieces of the reference file (EqualLR, AdaIN, binary_cross_entropy) are dead code
there and are not reproduced.
"""
import math

import torch
import torch.nn as nn

from meshvae_hip import functional as F_hip
from meshvae_hip import topology
from nn.conv import ChebConv_batch
from nn.pool import SurfacePool

# log_sigma = softclip(1, -6) = -6 + softplus(7): a constant (cheb_VAE.py:329-330, logpdf.py:24-28)
LOG_SIGMA = float(torch.nn.functional.softplus(torch.tensor([7.0])) - 6.0)


class _FusedModelFn(torch.autograd.Function):
    """cheb_VAE.forward as one autograd node (see cheb_VAE._forward_fused).  Only `loss` is differentiable.

    What the reference's loop does around it -- optimizer.zero_grad(); loss.backward(); optimizer.step() (main.py:74-81)
    -- runs on the calling thread, so everything here is written for that thread's time: outputs are FRESH tensors the
    native step writes directly (no clones), the gradients are views of one fresh flat buffer (autograd adopts them as
    the parameters' .grad without a copy; a kept .grad is accumulated into, as for any node), nothing synchronises, and
    the launch sequences themselves go to the device's asynchronous launcher when there is one (csrc/launcher.hip)."""

    @staticmethod
    def forward(ctx, ent, lch, x, x_gt, y, eps, drop_u, *params):
        step = ent["step"]
        outs, y_f = ent["alloc"](x, x_gt, y)
        step.run_forward(x, x_gt, y_f, eps, drop_u, outs, lch)
        loss, correct, recon, kld, rec, z_, y_hat, mu, logvar = outs
        ctx.ent, ctx.gen, ctx.lch = ent, ent["gen"], lch
        ctx.saved = (x, x_gt, y_f, eps, drop_u, recon, y_hat, mu, logvar)
        ent["keep"].append(ctx.saved + (loss, correct, kld, rec, z_))
        res = (loss, correct, recon, kld, rec, z_, y_hat)
        ctx.mark_non_differentiable(*res[1:])
        ctx.set_materialize_grads(False)          # (no zero tensors for the six outputs nobody differentiates)
        return res

    @staticmethod
    def backward(ctx, d_loss, *_):
        ent = ctx.ent
        if ent["gen"] != ctx.gen:
            raise RuntimeError("cheb_VAE fused forward: backward() must follow its own forward (the activations of an "
                               "earlier forward of this batch size were overwritten); set net.fused_step = False "
                               "to keep several graphs alive")
        if d_loss is None:                         # (materialize_grads is off: loss did not reach the differentiated output)
            return (None,) * (7 + len(ent["views"]))
        x, x_gt, y_f, eps, drop_u, recon, y_hat, mu, logvar = ctx.saved
        step = ent["step"]
        d_loss = d_loss.contiguous()
        # a fresh flat gradient buffer per backward: autograd may keep what it is handed as .grad, and the caller may
        # keep THAT over any number of steps (no zero_grad, gradient accumulation): nothing here is ever rewritten
        flat = torch.empty(ent["numel"], dtype=torch.float32, device=x.device)
        ent["G_np"][:] = ent["off_np"] + flat.data_ptr()
        step.run_backward(x, x_gt, y_f, eps, drop_u, d_loss, recon, y_hat, mu, logvar, ent["G"], ctx.lch)
        ent["keep"].append(ctx.saved + (flat, d_loss))
        as_strided = flat.as_strided
        # (dec_lin_1 is never used by the forward, reference :165: its gradient is None, its span of `flat` is not written)
        return (None, None, None, None, None, None, None,
                *[None if v is None else as_strided(v[0], v[1], v[2]) for v in ent["views"]])


class _FusedModelAssignFn(torch.autograd.Function):
    """net.grad_mode = "assign" (opt-in): the same fused step, but autograd sees ONE differentiable input (a dummy scalar
    of the module) instead of the 31 parameters, and the backward ASSIGNS every parameter's .grad itself -- views of a
    flat buffer, two buffers in rotation -- instead of returning 29 gradients to 29 AccumulateGrad nodes (about 100 us
    of the ~150 us `loss.backward()` costs the calling thread).  What it gives up: gradient ACCUMULATION over several
    backward calls, parameter hooks, torch.autograd.grad(loss, params), DDP wrappers -- the plain main.py loop
    (zero_grad -> net() -> backward -> optimizer.step, main.py:74-81) uses none of them."""

    @staticmethod
    def forward(ctx, ent, lch, x, x_gt, y, eps, drop_u, dummy):
        step = ent["step"]
        outs, y_f = ent["alloc"](x, x_gt, y)
        step.run_forward(x, x_gt, y_f, eps, drop_u, outs, lch)
        loss, correct, recon, kld, rec, z_, y_hat, mu, logvar = outs
        ctx.ent, ctx.gen, ctx.lch = ent, ent["gen"], lch
        ctx.saved = (x, x_gt, y_f, eps, drop_u, recon, y_hat, mu, logvar)
        ent["keep"].append(ctx.saved + (loss, correct, kld, rec, z_))
        res = (loss, correct, recon, kld, rec, z_, y_hat)
        ctx.mark_non_differentiable(*res[1:])
        ctx.set_materialize_grads(False)
        return res

    @staticmethod
    def backward(ctx, d_loss, *_):
        ent = ctx.ent
        if ent["gen"] != ctx.gen:
            raise RuntimeError("cheb_VAE fused forward: backward() must follow its own forward")
        if d_loss is None:
            return (None,) * 8
        x, x_gt, y_f, eps, drop_u, recon, y_hat, mu, logvar = ctx.saved
        step = ent["step"]
        d_loss = d_loss.contiguous()
        slot = ent["assign"][ent["assign_i"]]
        ent["assign_i"] ^= 1
        flat, G, grads = slot
        step.run_backward(x, x_gt, y_f, eps, drop_u, d_loss, recon, y_hat, mu, logvar, G, ctx.lch)
        ent["keep"].append(ctx.saved + (d_loss,))
        for p, g in zip(ent["params"], grads):
            if g is not None:
                p.grad = g
        return (None,) * 8


class cheb_VAE(torch.nn.Module):

    def __init__(self, num_features, config, downsample_matrices, upsample_matrices,
                 adjacency_matrices, num_nodes, model='MSE_VAE'):
        super().__init__()
        self.n_layers = config['n_layers']
        self.filters = [num_features] + list(config['num_conv_filters'])
        self.K = config['polygon_order']
        self.downsample_matrices = downsample_matrices
        self.upsample_matrices = upsample_matrices
        self.adjacency_matrices = adjacency_matrices
        levels = [ChebConv_batch.norm(adjacency_matrices[i]._indices(), num_nodes[i])
                  for i in range(len(num_nodes))]
        self.A_edge_index, self.A_norm = zip(*levels)
        self.num_nodes = list(num_nodes)

        f = self.filters
        # creation order == RNG order of the reference (encoder convs, decoder convs, linears)
        self.cheb = nn.ModuleList(ChebConv_batch(f[i], f[i + 1], self.K[i]) for i in range(len(f) - 2))
        self.cheb_dec = nn.ModuleList(ChebConv_batch(f[-i - 1], f[-i - 2], self.K[i]) for i in range(len(f) - 1))
        self.cheb_dec[-1].bias = None          # last conv has no bias (its draw is discarded)
        self.pool = SurfacePool()

        self.num_class = config['num_classes']
        self.z = config['num_style']
        self.num_hidden = config['num_hidden']
        flat = self.downsample_matrices[-1].shape[0] * f[-1]
        self.classifier_layer = nn.Linear(self.num_hidden, self.num_class)
        self.z_mean = nn.Linear(self.num_hidden + self.num_class, self.z)
        self.z_log_var = nn.Linear(self.num_hidden + self.num_class, self.z)
        self.enc_lin = nn.Linear(flat, self.num_hidden)
        self.dec_lin = nn.Linear(self.z + self.num_class, self.num_hidden)
        self.dec_lin_1 = nn.Linear(self.z + self.num_class, self.num_hidden)   # unused, kept for state_dict
        self.dec_lin_2 = nn.Linear(self.num_hidden, flat)
        self.dropout = nn.Dropout(p=config['dropout'])
        self.reset_parameters()
        self.type = model
        self._ops_ready = False

    # ------------------------------------------------------------------ topology
    def _prepare(self):
        """Device CSR of every A / D / U level (built once; tensors are immutable inputs)."""
        if self._ops_ready:
            return
        n = self.n_layers
        self._lap = [topology.laplacian(self.A_edge_index[i], self.A_norm[i], self.num_nodes[i]) for i in range(n)]
        # final layer: coarsest edge list applied to the finest vertex set (reference :288)
        self._lap_final = topology.laplacian(self.A_edge_index[-1], self.A_norm[-1], self.num_nodes[0])
        self._down = [topology.pool_operator(m) for m in self.downsample_matrices]
        self._up = [topology.pool_operator(m) for m in self.upsample_matrices]
        self._ops_ready = True

    def _fused_names(self):
        return [n for n, _ in self.named_parameters()]

    def reset_parameters(self):
        nn.init.normal_(self.enc_lin.weight, 0, 0.1)
        nn.init.normal_(self.dec_lin.weight, 0, 0.1)

    def set_param(self, alpha, beta):
        self.alpha = alpha
        self.beta = beta

    # ------------------------------------------------------------------ dropout plumbing
    def _drop_u(self, rows, cols, device):
        """Uniform randoms for one dropout site, or None in eval mode / p == 0."""
        if not self.training or self.dropout.p <= 0.0:
            return None
        return torch.rand(rows, cols, device=device, dtype=torch.float32)

    # ------------------------------------------------------------------ stages
    def encoder(self, x):
        self._prepare()
        if x.dim() == 3 and self._native_ok(x):
            ent = self._fused_entry(x.shape[0], x.device)
            ent["gen"] += 1                          # (the step's workspace is reused: invalidates a pending fused backward)
            return ent["step"].encode(x, self._drop_u(x.shape[0], self.num_hidden, x.device))
        for i in range(self.n_layers):
            x = F_hip.cheb_conv(x, self.cheb[i].weight, self.cheb[i].bias, self._lap[i], relu=True)
            x = F_hip.surface_pool(x, self._down[i])
        x = x.reshape(x.shape[0], self.enc_lin.in_features)
        return F_hip.linear(x, self.enc_lin.weight, self.enc_lin.bias, relu=True,
                            drop_u=self._drop_u(x.shape[0], self.num_hidden, x.device), p=self.dropout.p)

    def _latent(self, h, y, m_type):
        """classifier + latent heads + reparameterisation in one fused kernel."""
        B = h.shape[0]
        eps = None
        if m_type == "train":
            provider = getattr(self, "_eps_provider", None)
            if provider is not None:       # engine.TrainStep: static device buffer refilled from the host RNG
                eps = provider(B, self.z, h.device)     # (None for a batch size the buffer was not built for)
            if eps is None:
                # drawn on the host default generator, exactly as the reference does (:316)
                eps = torch.normal(mean=0, std=1, size=(B, self.z)).to(h.device)
        return F_hip.latent_head(h, y.to(torch.float32), self.classifier_layer.weight, self.classifier_layer.bias,
                                 self.z_mean.weight, self.z_mean.bias, self.z_log_var.weight, self.z_log_var.bias,
                                 drop_u=self._drop_u(B, self.num_hidden, h.device), p=self.dropout.p, eps=eps)

    def classifier(self, x):
        B = x.shape[0]
        y0 = torch.zeros(B, self.num_class, dtype=torch.float32, device=x.device)
        return self._latent(x, y0, "test")[0]

    def decoder(self, x):
        self._prepare()
        if x.dim() == 2 and self._native_ok(x):
            ent = self._fused_entry(x.shape[0], x.device)
            ent["gen"] += 1
            step, B_, H = ent["step"], x.shape[0], self.num_hidden
            u = None
            if self.training and self.dropout.p > 0.0:     # same draws, in the same order, as the per-module path
                u = torch.empty(B_ * step.u_cols, device=x.device)
                u[2 * B_ * H:3 * B_ * H] = torch.rand(B_, H, device=x.device).reshape(-1)
                u[3 * B_ * H:] = torch.rand(B_, self.dec_lin_2.out_features, device=x.device).reshape(-1)
            return ent["step"].decode(x, u)
        B, dev, p = x.shape[0], x.device, self.dropout.p
        x = F_hip.linear(x, self.dec_lin.weight, self.dec_lin.bias, relu=True,
                         drop_u=self._drop_u(B, self.num_hidden, dev), p=p)
        x = F_hip.linear(x, self.dec_lin_2.weight, self.dec_lin_2.bias, relu=True,
                         drop_u=self._drop_u(B, self.dec_lin_2.out_features, dev), p=p)
        x = x.reshape(B, -1, self.filters[-1])
        for i in range(self.n_layers):
            x = F_hip.surface_pool(x, self._up[-i - 1])
            conv = self.cheb_dec[i]
            x = F_hip.cheb_conv(x, conv.weight, conv.bias, self._lap[self.n_layers - i - 1], relu=True)
        last = self.cheb_dec[-1]
        return F_hip.cheb_conv(x, last.weight, last.bias, self._lap_final, relu=False)

    def sample(self, y, z):
        zy = torch.cat([y.to(z.dtype), z], -1)
        return self.decoder(zy).reshape(z.shape[0], -1, self.filters[0])

    def reparameterize(self, mu, logvar):
        eps = torch.normal(mean=0, std=1, size=(mu.shape[0], logvar.shape[1])).to(mu.device)
        return eps * torch.exp(logvar * 0.5) + mu

    def loss_function(self, x, recon_x, z, mu_z, logvar_z, y, y_hat):
        return F_hip.vae_loss(recon_x, x, mu_z, logvar_z, y.to(torch.float32), y_hat, LOG_SIGMA)

    # ------------------------------------------------------------------ full step
    def _host_eps(self, B, dev):
        """The reparameterisation noise exactly as the reference draws it -- torch.normal on the HOST default generator
        (:316) -- but moved with an asynchronous copy from a small ring of pinned buffers: the reference's pageable
        `.to(device)` makes the host wait for everything queued on the stream, once per step."""
        rings = self.__dict__.setdefault("_eps_rings", {})
        key = (B, dev.index)
        ring = rings.get(key)
        if ring is None:
            if len(rings) >= 4:
                rings.pop(next(iter(rings)))
            ring = rings[key] = {"i": 0, "slots": [(torch.empty(B, self.z).pin_memory(), torch.cuda.Event()) for _ in range(4)]}
        buf, ev = ring["slots"][ring["i"]]
        ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        ev.synchronize()                                   # the copy that read this buffer four draws ago
        torch.normal(mean=0, std=1, size=(B, self.z), out=buf)
        eps = torch.empty(B, self.z, device=dev)
        eps.copy_(buf, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        return eps

    def _forward_fused(self, x, x_gt, y, m_type):
        """The whole forward as ONE autograd node over the native step (mvh_vae_forward / mvh_vae_backward):
        what `loss.backward()` in the reference's train loop (main.py:80) then triggers is a single C++ launch
        sequence instead of ~60 Python-driven autograd nodes -- bitwise the same numbers (same kernels)."""
        B, dev = x.shape[0], x.device
        ent = self._fused_entry(B, dev)
        step = ent["step"]
        eps = None
        if m_type == "train":
            provider = getattr(self, "_eps_provider", None)
            eps = provider(B, self.z, dev) if provider is not None else None
            if eps is None:                # no engine buffer for this batch size: host generator, as the reference (:316)
                eps = self._host_eps(B, dev)
        drop_u = None
        if self.training and self.dropout.p > 0.0:
            # the uniforms of U_AHEAD calls from ONE generator launch (as engine.TrainStep does): a call takes its row
            ring = ent["drop"]
            if ring["left"] == 0:
                ring["i"] ^= 1
                ring["buf"][ring["i"]] = torch.rand(self.U_AHEAD, B * step.u_cols, device=dev)   # (a fresh block: earlier rows may still be in use)
                ring["left"] = self.U_AHEAD
            drop_u = ring["buf"][ring["i"]][self.U_AHEAD - ring["left"]]
            ring["left"] -= 1
        ent["gen"] += 1
        x, x_gt = x.contiguous(), x_gt.contiguous()
        # (inside a stream capture -- TrainStep(native=False, use_graph=True) -- the launches must be recorded by the
        #  capturing thread on the capturing stream: no launcher there)
        lch = ent["launcher"] if (ent["launcher"] is not None and not torch.cuda.is_current_stream_capturing()) else None
        if not torch.is_grad_enabled():              # evaluate loops (main.py:129): the same launch sequence, forward only
            outs, y_f = ent["alloc"](x, x_gt, y)
            step.run_forward(x, x_gt, y_f, eps, drop_u, outs, lch)
            ent["keep"].append((x, x_gt, y_f, eps, drop_u) + outs)
            loss, correct, recon, kld, rec, z_, y_hat = outs[:7]
            return loss, correct, recon, [kld, rec, z_], y_hat
        if getattr(self, "grad_mode", "autograd") == "assign":
            outs = _FusedModelAssignFn.apply(ent, lch, x, x_gt, y, eps, drop_u, self._assign_dummy(dev))
        else:
            outs = _FusedModelFn.apply(ent, lch, x, x_gt, y, eps, drop_u, *step.params)
        loss, correct, recon, kld, rec, z_, y_hat = outs
        return loss, correct, recon, [kld, rec, z_], y_hat

    def _fused_entry(self, B, dev):
        """Native step object (descriptor, workspace) for batch size B plus what the autograd node needs per call, a few
        sizes cached."""
        import collections
        import ctypes

        import numpy as np

        import meshvae_hip
        from meshvae_hip.engine import NativeStep
        cache = self.__dict__.setdefault("_fused_cache", {})
        # net.storage = "bf16": the fused forward / backward keep the activations between the conv layers as bf16
        # (mvh_vae_desc_t.storage; fp32 accumulation, fp32 parameters) -- BASELINE configs[1] as worded.  Default fp32.
        storage = getattr(self, "storage", "f32")
        B_key = (B, storage, dev.index)
        ent = cache.get(B_key)
        if ent is None:
            if len(cache) >= 3:                      # (a workspace is ~12 MB per mesh: keep few batch sizes)
                old = cache.pop(next(iter(cache)))
                if old["launcher"] is not None:      # (its workspace may still be named by a queued job)
                    meshvae_hip.check(meshvae_hip.lib().mvh_launcher_sync(old["launcher"]))
            names = self._fused_names()
            params = [p for _, p in self.named_parameters()]
            # gradients of one backward: views of ONE flat fp32 buffer, every tensor on a 256-byte boundary
            offs, off = [], 0
            for p in params:
                offs.append(off)
                off += -(-p.numel() // 64) * 64
            views = [None if n.startswith("dec_lin_1.") else (tuple(p.shape), tuple(p.stride()), o)
                     for n, p, o in zip(names, params, offs)]
            # the step's own gradient table is unused on this path (run_backward gets `G`): no per-parameter buffers
            step = NativeStep(self, B, grads="external", storage=storage)
            G = (ctypes.c_void_p * len(params))()
            N0, F0, Z, C = self.num_nodes[0], self.filters[0], self.z, self.num_class

            def alloc(x, x_gt, y, B=B, dev=dev, N0=N0, F0=F0, Z=Z, C=C):
                """Fresh output tensors of one forward (written by the native step) and y as the fp32 one-hot it reads."""
                f32 = {"dtype": torch.float32, "device": dev}
                y_f = y if (y.dtype == torch.float32 and y.is_contiguous()) else y.to(torch.float32).contiguous()
                lt = x_gt.dtype
                return ((torch.empty((), dtype=lt, device=dev), torch.empty((), dtype=torch.int64, device=dev),
                         torch.empty(B, N0, F0, **f32), torch.empty(B, **f32), torch.empty(B, dtype=lt, device=dev),
                         torch.empty(B, Z, **f32), torch.empty(B, C, **f32), torch.empty(B, Z, **f32),
                         torch.empty(B, Z, **f32)), y_f)
            def assign_slot():
                """one flat gradient buffer of the "assign" mode with its pointer table and the parameters' views of it"""
                flat = torch.empty(off, dtype=torch.float32, device=dev)
                Gs = (ctypes.c_void_p * len(params))()
                np.frombuffer(Gs, dtype=np.uint64)[:] = np.asarray(offs, dtype=np.uint64) * np.uint64(4) + np.uint64(flat.data_ptr())
                grads = [None if v is None else flat.as_strided(v[0], v[1], v[2]) for v in views]
                return flat, Gs, grads
            ent = cache[B_key] = {"step": step, "gen": 0, "numel": off, "views": views, "G": G,
                                  "params": params, "assign": None, "assign_i": 0, "assign_slot": assign_slot,
                                  "drop": {"left": 0, "i": 0, "buf": [None, None]},
                                  "G_np": np.frombuffer(G, dtype=np.uint64),
                                  "off_np": np.asarray(offs, dtype=np.uint64) * np.uint64(4), "alloc": alloc,
                                  # the tensors of the last few calls stay referenced: a job handed to the asynchronous
                                  # launcher names raw addresses, and the caller may drop its tensors (or empty the
                                  # allocator's cache) before the worker thread has enqueued it
                                  "keep": collections.deque(maxlen=6),
                                  "launcher": meshvae_hip.launcher(dev.index) if getattr(self, "async_launch", True) else None}
        ent["step"].refresh_param_pointers()
        if getattr(self, "grad_mode", "autograd") == "assign" and ent["assign"] is None:
            ent["assign"] = [ent["assign_slot"](), ent["assign_slot"]()]
        return ent

    U_AHEAD = 16      # calls whose dropout uniforms one generator launch draws (module path; engine.TrainStep has its own)

    def _assign_dummy(self, dev):
        d = self.__dict__.get("_assign_dummy_t")
        if d is None or d.device != dev:
            d = self.__dict__["_assign_dummy_t"] = torch.zeros((), device=dev, requires_grad=True)
        return d

    def _native_ok(self, t):
        """Piecewise inference calls (net.encoder / net.decoder under no_grad) take the native launch sequences."""
        return getattr(self, "fused_step", True) and t.is_cuda and not torch.is_grad_enabled() and t.dtype == torch.float32

    def forward(self, data, x_gt, y, supervise=True, m_type="test"):
        self.supervise = supervise
        x, batch_size = data.x, data.num_graphs          # data.edge_index is never used (reference :195)
        x = x.reshape(batch_size, -1, self.filters[0])
        if (getattr(self, "fused_step", True) and x.is_cuda and x_gt.dtype in (torch.float32, torch.float64)
                and (not torch.is_grad_enabled() or (not x.requires_grad and not x_gt.requires_grad
                                                     and self._all_params_require_grad()))):
            self._prepare()
            return self._forward_fused(x, x_gt, y, m_type)
        h = self.encoder(x)
        y_hat, x_mean, x_var, z_, z = self._latent(h, y, m_type)
        x = self.decoder(z).reshape(batch_size, -1, self.filters[0])
        loss, correct, kld, rec_loss = self.loss_function(x_gt, x, z, x_mean, x_var, y, y_hat)
        return loss, correct, x, [kld, rec_loss, z_], y_hat

    def _all_params_require_grad(self):
        """(over a cached list: walking the module tree with .parameters() costs ~40 us per call.  The list holds, per
        parameter, the owning module's _parameters dict and its key: a Parameter OBJECT that was replaced since --
        `net.cls.weight = nn.Parameter(...)`, parametrizations, conversions under
        torch.__future__.set_overwrite_module_params_on_conversion -- is seen by identity, and every cache that names the
        old objects (this list, the fused entries with their gradient views) is dropped and rebuilt.)"""
        ps = self.__dict__.get("_param_list")
        if ps is None:
            ps = self.__dict__["_param_list"] = self._param_triples()
        for d, n, p in ps:
            if d.get(n) is not p:
                self._drop_param_caches()
                return self._all_params_require_grad()
            if not p.requires_grad:
                return False
        return True

    def _param_triples(self):
        out, seen = [], set()
        for m in self.modules():
            for n, p in m._parameters.items():
                if p is not None and id(p) not in seen:      # (named_parameters' order and de-duplication)
                    seen.add(id(p))
                    out.append((m._parameters, n, p))
        return out

    def _drop_param_caches(self):
        import meshvae_hip
        self.__dict__.pop("_param_list", None)
        for ent in self.__dict__.pop("_fused_cache", {}).values():
            if ent["launcher"] is not None:          # (its workspace may still be named by a queued job)
                meshvae_hip.check(meshvae_hip.lib().mvh_launcher_sync(ent["launcher"]))
