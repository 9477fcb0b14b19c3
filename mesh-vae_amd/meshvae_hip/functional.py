"""torch.autograd.Function wrappers over the C ABI (include/meshvae_hip.h).

Each Function launches hand-written HIP kernels on the CURRENT stream of the tensors'
device (so the autograd engine thread and hipGraph capture both work) and implements the
analytic backward the reference leaves to autograd.  No torch arithmetic happens here.
"""
import torch

from . import MeshVaeHipUnsupported, check, lib

_ws = {}


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mesh-vae_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback)")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32 tensor, got {t.dtype}")


def _c(t):
    return None if t is None else t.contiguous()


def _row_view(x, channels_multiple=True):
    """(mesh stride, row stride) in elements if the [B, N, C] tensor x can cross the boundary WITHOUT a copy through the
    strided entry points (include/meshvae_hip.h: channel stride 1, positive strides that are multiples of C, 16-byte
    aligned rows), None if it is contiguous already or does not qualify (the caller then makes the contiguous copy).
    The reference's own modules produce exactly such views: x.transpose(0, 1) of an [N, B, C] tensor (nn/conv.py:560)."""
    if x.dim() != 3 or x.is_contiguous():
        return None
    B, N, C = x.shape
    sb, sv, sc = x.stride()
    if sc != 1 or sb <= 0 or sv <= 0 or B == 0 or N == 0:
        return None
    if channels_multiple and (sb % C or sv % C):
        return None
    if x.data_ptr() % 16 or (C % 4 == 0 and (sb % 4 or sv % 4)):
        return None
    return sb, sv


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream); kernels using it are stream-ordered."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


# ----------------------------------------------------------------------------- row S
class PoolFn(torch.autograd.Function):
    """SurfacePool.forward (nn/pool.py:17-20) and its backward."""

    @staticmethod
    def forward(ctx, x, op):
        _need_gpu(x)
        if x.dim() != 3:
            raise ValueError("SurfacePool expects x of shape [B, N, C]")
        view = _row_view(x, channels_multiple=False)      # e.g. the transpose of an [N, B, C] tensor: read in place
        if view is None:
            x = x.contiguous()
        B, N, C = x.shape
        if N != op.n_in:
            # same check and message as MessagePassing.__set_size__ (nn/conv.py:165-169)
            raise ValueError(f"Encountered node tensor with size {N} in dimension 0, "
                             f"but expected size {op.n_in}.")
        y = torch.empty(B, op.n_out, C, dtype=x.dtype, device=x.device)
        ctx.op = op
        if B == 0:
            return y
        with torch.cuda.device(x.device):
            if view is not None:
                check(lib().mvh_pool_fwd_strided(_stream(x), op.fwd.ref, x.data_ptr(), view[0], view[1], y.data_ptr(), B, C))
            else:
                check(lib().mvh_pool_fwd(_stream(x), op.fwd.ref, x.data_ptr(), y.data_ptr(), B, C))
        ctx.op = op
        return y

    @staticmethod
    def backward(ctx, dy):
        op = ctx.op
        dy = dy.contiguous()
        B, _, C = dy.shape
        dx = torch.empty(B, op.n_in, C, dtype=dy.dtype, device=dy.device)
        if B == 0:
            return dx, None
        with torch.cuda.device(dy.device):
            check(lib().mvh_pool_bwd(_stream(dy), op.bwd.ref, dy.data_ptr(), dx.data_ptr(), B, C))
        return dx, None


# ----------------------------------------------------------------------------- rows C + Q
class ChebConvFn(torch.autograd.Function):
    """ChebConv_batch.forward (nn/conv.py:557-577), optionally fused with F.relu."""

    @staticmethod
    def forward(ctx, x, weight, bias, op, act):
        _need_gpu(x, weight, bias)
        if x.dim() != 3:
            raise ValueError("ChebConv_batch expects x of shape [B, N, C_in]")
        weight, bias = weight.contiguous(), _c(bias)
        B, N, Cin = x.shape
        K, Cin_w, Cout = weight.shape
        if Cin != Cin_w:
            raise ValueError(f"ChebConv_batch: x has {Cin} channels but weight expects {Cin_w}")
        if N != op.n_out:
            raise ValueError(f"Encountered node tensor with size {N} in dimension 0, "
                             f"but expected size {op.n_out}.")
        out = torch.empty(B, N, Cout, dtype=x.dtype, device=x.device)
        L = lib()
        signs = None
        if act and Cout % 4 == 0 and K > 1:
            # fused ReLU: keep its signs as one byte per 4 channels, the backward reads those
            signs = torch.empty(B, N, Cout // 4, dtype=torch.uint8, device=x.device)
        # A non-contiguous x whose rows are intact (x.transpose(0, 1) of an [N, B, C] tensor, a batch slice, ...) is read
        # in place by the LDS-resident kernels; shapes they do not cover answer "unsupported" and take the copy below.
        view = _row_view(x) if K > 1 else None
        if B == 0:                              # an empty batch (empty tensors have no address): nothing to launch
            ctx.op, ctx.act, ctx.has_bias, ctx.view = op, act, bias is not None, None
            ctx.save_for_backward(x, weight, out if act else None, None, signs)
            return out
        if view is not None:
            with torch.cuda.device(x.device):
                ws_bytes = L.mvh_cheb_conv_strided_ws_bytes(B, N, Cin, Cout, K)
                ws = workspace(ws_bytes, x.device)
                try:
                    check(L.mvh_cheb_conv_fwd_strided(_stream(x), op.fwd.ref, x.data_ptr(), view[0], view[1], weight.data_ptr(),
                                                      _ptr(bias), out.data_ptr(), _ptr(signs), B, N, Cin, Cout, K, act,
                                                      ws.data_ptr(), ws_bytes))
                except MeshVaeHipUnsupported:
                    view = None
        if view is None:
            x = x.contiguous()
            with torch.cuda.device(x.device):
                ws = None
                ws_bytes = 0
                if K > 1:
                    # no T_k stack is saved: the fused kernels recompute the recurrence on chip
                    ws_bytes = L.mvh_cheb_conv_ws_bytes(B, N, Cin, Cout, K)
                    ws = workspace(ws_bytes, x.device)
                if signs is not None:
                    check(L.mvh_cheb_conv_fwd_signs(_stream(x), op.fwd.ref, x.data_ptr(), weight.data_ptr(), _ptr(bias),
                                                    out.data_ptr(), signs.data_ptr(), B, N, Cin, Cout, K, _ptr(ws), ws_bytes))
                else:
                    check(L.mvh_cheb_conv_fwd(_stream(x), op.fwd.ref, x.data_ptr(), weight.data_ptr(), _ptr(bias),
                                              out.data_ptr(), None, B, N, Cin, Cout, K, act, _ptr(ws), ws_bytes))
        ctx.op, ctx.act, ctx.has_bias, ctx.view = op, act, bias is not None, view
        ctx.save_for_backward(x, weight, out if act else None, None, signs)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out, tx, signs = ctx.saved_tensors
        op, act = ctx.op, ctx.act
        dout = dout.contiguous()
        B, N, Cin = x.shape
        K, _, Cout = weight.shape
        dx = torch.empty(B, N, Cin, dtype=x.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(Cout, dtype=x.dtype, device=x.device) if ctx.has_bias else None
        L = lib()
        view = ctx.view
        if B == 0:                              # empty batch: empty dx, zero parameter gradients (a sum over no meshes)
            return dx, dW.zero_(), (db.zero_() if db is not None else None), None, None
        if view is not None:                    # the saved x is the caller's strided view: the dW kernel reads it in place
            with torch.cuda.device(x.device):
                ws_bytes = L.mvh_cheb_conv_strided_ws_bytes(B, N, Cin, Cout, K)
                ws = workspace(ws_bytes, x.device)
                try:
                    check(L.mvh_cheb_conv_bwd_strided(_stream(x), op.fwd.ref, op.bwd.ref, x.data_ptr(), view[0], view[1],
                                                      weight.data_ptr(), _ptr(out), _ptr(signs), dout.data_ptr(), _ptr(dx),
                                                      dW.data_ptr(), _ptr(db), B, N, Cin, Cout, K, act, ws.data_ptr(), ws_bytes))
                except MeshVaeHipUnsupported:
                    view = None
        if view is None:
            x = x.contiguous()
            with torch.cuda.device(x.device):
                ws_bytes = L.mvh_cheb_conv_bwd_ws_bytes(B, N, Cin, Cout, K)
                ws = workspace(ws_bytes, x.device)
                if signs is not None:
                    check(L.mvh_cheb_conv_bwd_signs(_stream(x), op.fwd.ref, op.bwd.ref, x.data_ptr(), weight.data_ptr(),
                                                    out.data_ptr(), signs.data_ptr(), dout.data_ptr(), _ptr(dx),
                                                    dW.data_ptr(), _ptr(db), B, N, Cin, Cout, K, ws.data_ptr(), ws_bytes))
                else:
                    check(L.mvh_cheb_conv_bwd(_stream(x), op.fwd.ref, op.bwd.ref, x.data_ptr(), weight.data_ptr(),
                                              _ptr(out), dout.data_ptr(), _ptr(tx), _ptr(dx), dW.data_ptr(), _ptr(db),
                                              B, N, Cin, Cout, K, act, ws.data_ptr(), ws_bytes))
        return dx, dW, db, None, None


# ----------------------------------------------------------------------------- nn.Linear (+relu +dropout)
class LinearFn(torch.autograd.Function):
    """nn.Linear -> F.relu -> nn.Dropout as used at cheb_VAE.py:270-272 and :277-280."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, drop_u, p):
        _need_gpu(x, weight, bias, drop_u)
        x, weight, bias, drop_u = x.contiguous(), weight.contiguous(), _c(bias), _c(drop_u)
        B, fin = x.shape
        fout = weight.shape[0]
        if weight.shape[1] != fin:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{fin} and {weight.shape[1]}x{fout})")
        y = torch.empty(B, fout, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            check(lib().mvh_linear_fwd(_stream(x), x.data_ptr(), weight.data_ptr(), _ptr(bias), y.data_ptr(),
                                       B, fin, fout, act, _ptr(drop_u), float(p if drop_u is not None else 0.0)))
        ctx.act, ctx.p, ctx.has_bias = act, float(p if drop_u is not None else 0.0), bias is not None
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        B, fin = x.shape
        fout = weight.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(fout, dtype=x.dtype, device=x.device) if ctx.has_bias else None
        with torch.cuda.device(x.device):
            ws_bytes = B * fout * 4
            ws = workspace(ws_bytes, x.device)
            check(lib().mvh_linear_bwd(_stream(x), x.data_ptr(), weight.data_ptr(), y.data_ptr(), dy.data_ptr(),
                                       _ptr(dx), dW.data_ptr(), _ptr(db), B, fin, fout, ctx.act, ctx.p,
                                       ws.data_ptr(), ws_bytes))
        return dx, dW, db, None, None, None


# ----------------------------------------------------------------------------- rows K + Z + R
class LatentFn(torch.autograd.Function):
    """classifier + z_mean + z_log_var + reparameterize (cheb_VAE.py:203-226, 253-258, 309-319)."""

    @staticmethod
    def forward(ctx, h, y, Wc, bc, Wm, bm, Wv, bv, drop_u, p, eps):
        _need_gpu(h, y, Wc, bc, Wm, bm, Wv, bv, drop_u, eps)
        h, y = h.contiguous(), y.contiguous()
        Wc, bc, Wm, bm, Wv, bv = (t.contiguous() for t in (Wc, bc, Wm, bm, Wv, bv))
        drop_u, eps = _c(drop_u), _c(eps)
        B, H = h.shape
        C, Z = Wc.shape[0], Wm.shape[0]
        if Wm.shape[1] != H + C or Wv.shape != Wm.shape or Wc.shape[1] != H or y.shape != (B, C):
            raise RuntimeError("latent head: inconsistent shapes")
        new = lambda n: torch.empty(B, n, dtype=h.dtype, device=h.device)  # noqa: E731
        y_hat, mu, logvar, z, zy = new(C), new(Z), new(Z), new(Z), new(C + Z)
        p = float(p if drop_u is not None else 0.0)
        with torch.cuda.device(h.device):
            check(lib().mvh_vae_latent_fwd(_stream(h), h.data_ptr(), y.data_ptr(), _ptr(drop_u), p,
                                           Wc.data_ptr(), bc.data_ptr(), Wm.data_ptr(), bm.data_ptr(),
                                           Wv.data_ptr(), bv.data_ptr(), _ptr(eps), y_hat.data_ptr(),
                                           mu.data_ptr(), logvar.data_ptr(), z.data_ptr(), zy.data_ptr(),
                                           B, H, C, Z))
        ctx.p = p
        ctx.save_for_backward(h, y, Wc, Wm, Wv, drop_u, eps, y_hat, logvar)
        return y_hat, mu, logvar, z, zy

    @staticmethod
    def backward(ctx, d_yhat, d_mu, d_logvar, d_z, d_zy):
        h, y, Wc, Wm, Wv, drop_u, eps, y_hat, logvar = ctx.saved_tensors
        B, H = h.shape
        C, Z = Wc.shape[0], Wm.shape[0]
        d_zy = d_zy.contiguous().clone()
        d_zy[:, C:] += d_z                      # z_ is also returned on its own (cheb_VAE.py:251)
        d_yhat, d_mu, d_logvar = d_yhat.contiguous(), d_mu.contiguous(), d_logvar.contiguous()
        dh = torch.empty_like(h)
        dWc, dWm, dWv = torch.empty_like(Wc), torch.empty_like(Wm), torch.empty_like(Wv)
        dbc = torch.empty(C, dtype=h.dtype, device=h.device)
        dbm = torch.empty(Z, dtype=h.dtype, device=h.device)
        dbv = torch.empty(Z, dtype=h.dtype, device=h.device)
        with torch.cuda.device(h.device):
            ws_bytes = B * (C + 2 * Z) * 4
            ws = workspace(ws_bytes, h.device)
            check(lib().mvh_vae_latent_bwd(_stream(h), h.data_ptr(), y.data_ptr(), _ptr(drop_u), ctx.p,
                                           Wc.data_ptr(), Wm.data_ptr(), Wv.data_ptr(), _ptr(eps),
                                           y_hat.data_ptr(), logvar.data_ptr(), d_yhat.data_ptr(),
                                           d_mu.data_ptr(), d_logvar.data_ptr(), d_zy.data_ptr(),
                                           dh.data_ptr(), dWc.data_ptr(), dbc.data_ptr(), dWm.data_ptr(),
                                           dbm.data_ptr(), dWv.data_ptr(), dbv.data_ptr(), B, H, C, Z,
                                           ws.data_ptr(), ws_bytes))
        return dh, None, dWc, dbc, dWm, dbm, dWv, dbv, None, None, None


# ----------------------------------------------------------------------------- row L
class LossFn(torch.autograd.Function):
    """cheb_VAE.loss_function (cheb_VAE.py:321-346)."""

    @staticmethod
    def forward(ctx, recon, x_gt, mu, logvar, y, y_hat, log_sigma):
        _need_gpu(recon, mu, logvar, y, y_hat)
        if not x_gt.is_cuda:
            raise RuntimeError("mesh-vae_amd runs on MI355X only: x_gt is a CPU tensor")
        if x_gt.dtype not in (torch.float32, torch.float64):
            raise TypeError(f"x_gt must be float32 or float64, got {x_gt.dtype}")
        recon, x_gt = recon.contiguous(), x_gt.contiguous()
        mu, logvar, y, y_hat = mu.contiguous(), logvar.contiguous(), y.contiguous(), y_hat.contiguous()
        B = recon.shape[0]
        NV = recon[0].numel()
        if x_gt.numel() != recon.numel():
            raise RuntimeError(f"The size of tensor a ({x_gt.numel()}) must match the size of tensor b ({recon.numel()})")
        C, Z = y.shape[1], mu.shape[1]
        f64 = x_gt.dtype == torch.float64
        loss = torch.empty((), dtype=x_gt.dtype, device=recon.device)
        rec = torch.empty(B, dtype=x_gt.dtype, device=recon.device)
        kld = torch.empty(B, dtype=torch.float32, device=recon.device)
        correct = torch.empty((), dtype=torch.int64, device=recon.device)
        L = lib()
        with torch.cuda.device(recon.device):
            ws_bytes = L.mvh_vae_loss_ws_bytes(B)
            ws = workspace(ws_bytes, recon.device)
            check(L.mvh_vae_loss_fwd(_stream(recon), recon.data_ptr(), x_gt.data_ptr(), int(f64), mu.data_ptr(),
                                     logvar.data_ptr(), y.data_ptr(), y_hat.data_ptr(), float(log_sigma),
                                     loss.data_ptr(), rec.data_ptr(), kld.data_ptr(), correct.data_ptr(),
                                     B, NV, C, Z, ws.data_ptr(), ws_bytes))
        ctx.log_sigma, ctx.f64 = float(log_sigma), f64
        ctx.save_for_backward(recon, x_gt, mu, logvar, y, y_hat)
        ctx.mark_non_differentiable(correct, kld, rec)
        return loss, correct, kld, rec

    @staticmethod
    def backward(ctx, d_loss, _dc, _dk, _dr):
        recon, x_gt, mu, logvar, y, y_hat = ctx.saved_tensors
        B = recon.shape[0]
        NV = recon[0].numel()
        C, Z = y.shape[1], mu.shape[1]
        d_loss = d_loss.to(x_gt.dtype).contiguous()
        d_recon, d_mu, d_logvar = torch.empty_like(recon), torch.empty_like(mu), torch.empty_like(logvar)
        d_yhat = torch.empty_like(y_hat)
        with torch.cuda.device(recon.device):
            check(lib().mvh_vae_loss_bwd(_stream(recon), recon.data_ptr(), x_gt.data_ptr(), int(ctx.f64),
                                         mu.data_ptr(), logvar.data_ptr(), y.data_ptr(), y_hat.data_ptr(),
                                         ctx.log_sigma, d_loss.data_ptr(), d_recon.data_ptr(), d_mu.data_ptr(),
                                         d_logvar.data_ptr(), d_yhat.data_ptr(), B, NV, C, Z))
        return d_recon, None, d_mu, d_logvar, None, d_yhat, None


def cheb_conv(x, weight, bias, op, relu=False):
    return ChebConvFn.apply(x, weight, bias, op, 1 if relu else 0)


def surface_pool(x, op):
    return PoolFn.apply(x, op)


def linear(x, weight, bias, relu=False, drop_u=None, p=0.0):
    return LinearFn.apply(x, weight, bias, 1 if relu else 0, drop_u, p)


def latent_head(h, y, Wc, bc, Wm, bm, Wv, bv, drop_u=None, p=0.0, eps=None):
    return LatentFn.apply(h, y, Wc, bc, Wm, bm, Wv, bv, drop_u, p, eps)


def vae_loss(recon, x_gt, mu, logvar, y, y_hat, log_sigma):
    return LossFn.apply(recon, x_gt, mu, logvar, y, y_hat, log_sigma)
