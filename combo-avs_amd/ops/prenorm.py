"""Pre-norm residual step of a PVTv2 block on csrc/prenorm.hip: z = x + s[b] * r, y = LN(z) -> bf16 in one pass, and its
backward (d = LN'(dy) + dz; dx = d; dr = bf16(s * d)) in one pass.  Replaces, per LayerNorm of the backbone, a cast, the
stochastic-depth multiply, a mixed-dtype add and the LayerNorm kernels (backbone/pvtv2.py:162-175 under autocast).  The
LayerNorm's parameter gradients join the deferred grouped launch of ops.linear.deferred_dw like those of ops.layernorm."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from . import linear as _linear_mod

WIDTHS = (64, 128, 256, 320, 512)


def usable(x, C):
    return x.is_cuda and x.dtype == torch.float32 and C in WIDTHS and x.is_contiguous()


class _PreNorm(Function):
    """(x fp32 [B, N, C], r bf16 [B, N, C] or None, scale fp32 [B] or None) -> (z, y) with r, y alone without.
    y is bf16 unless out_fp32.  fanout = 2: y is returned twice (aliases of one buffer, one autograd output per consumer) so that
    the two consumers' gradients arrive separately and are summed inside the backward kernel, not by an accumulation kernel."""

    @staticmethod
    def forward(ctx, x, r, scale, weight, bias, eps, out_fp32, defer, fanout=1):
        B, N, C = x.shape
        rows = B * N
        x2 = x.reshape(rows, C)
        r2 = None
        if r is not None:
            r2 = r.reshape(rows, C)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
            if r2.dtype != torch.bfloat16:
                r2 = r2.to(torch.bfloat16)
        z = torch.empty_like(x2) if r2 is not None else None
        y = torch.empty(rows, C, device=x.device, dtype=torch.float32 if out_fp32 else torch.bfloat16)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_prenorm_forward(x2.data_ptr(), _lib.ptr(r2), _lib.ptr(scale), N, weight.data_ptr(), bias.data_ptr(), eps,
                                                    rows, C, _lib.ptr(z), y.data_ptr(), 0 if out_fp32 else 1, mean.data_ptr(),
                                                    rstd.data_ptr(), _lib.current_stream()), "combo_prenorm_forward")
        ctx.save_for_backward(z if z is not None else x2, mean, rstd, weight, scale)
        ctx.has_r, ctx.defer, ctx.shape, ctx.N, ctx.fanout = r is not None, defer, x.shape, N, fanout
        y = y.view(x.shape)
        ys = (y,) if fanout == 1 else (y, y.view_as(y))
        if r is None:
            return ys[0] if fanout == 1 else ys
        return (z.view(x.shape),) + ys

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        z, mean, rstd, weight, scale = ctx.saved_tensors
        dz = grads[0] if ctx.has_r else None
        dys = [g for g in (grads[1:] if ctx.has_r else grads) if g is not None]
        rows, C = z.shape
        if dz is None and not dys:
            return (None,) * 9

        def flat(g):
            g = g.reshape(rows, C)
            return g if g.is_contiguous() else g.contiguous()
        dy = dy2 = None
        if dys:
            dys = [flat(g) for g in dys]
            if len({g.dtype for g in dys}) > 1 or dys[0].dtype not in (torch.bfloat16, torch.float32):
                dys = [g.float() for g in dys]
            dy, dy2 = dys[0], (dys[1] if len(dys) > 1 else None)
        if dz is not None:
            dz = dz.reshape(rows, C)
            dz = dz if dz.is_contiguous() else dz.contiguous()
        want_param = dy is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        dy32 = torch.empty(rows, C, device=z.device, dtype=torch.float32) \
            if (want_param and (dy.dtype != torch.float32 or dy2 is not None)) else None
        dx = torch.empty(rows, C, device=z.device, dtype=torch.float32)
        dr = torch.empty(rows, C, device=z.device, dtype=torch.bfloat16) if (ctx.has_r and ctx.needs_input_grad[1]) else None
        _lib.check(_lib.lib().combo_prenorm_backward(_lib.ptr(dy), _lib.ptr(dy2), 1 if (dy is not None and dy.dtype == torch.bfloat16) else 0, _lib.ptr(dz),
                                                     z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), weight.data_ptr(), _lib.ptr(scale),
                                                     ctx.N, rows, C, dx.data_ptr(), _lib.ptr(dr), _lib.ptr(dy32), _lib.current_stream()),
                   "combo_prenorm_backward")
        dw = db = None
        if want_param:
            use = (dy32 if dy32 is not None else dy, z, mean, rstd)
            q = _linear_mod._ln_queue
            out = torch.empty(2, C, device=z.device, dtype=torch.float32)
            if ctx.defer and q is not None:
                q.append([[use], out])  # filled when deferred_dw() closes (every LayerNorm of the backbone is applied once)
            else:
                _linear_mod._flush_ln([[[use], out]])
            dw, db = out[0], out[1]
        return (dx.view(ctx.shape), None if dr is None else dr.view(ctx.shape), None, dw, db, None, None, None, None)


def prenorm(x, r, scale, norm, out_fp32=False, defer=True, fanout=1):
    """norm: an nn.LayerNorm; returns (z, y[, y alias]) when r is given, y[, y alias] otherwise"""
    return _PreNorm.apply(x, r, scale, norm.weight, norm.bias, norm.eps, out_fp32, defer, fanout)


class _BiasLn(Function):
    """y = LN(x + xb[c]) on bf16 rows, bf16 out (the key / value path of a spatial-reduction attention block: the strided
    convolution runs without its bias, which is added here; its gradient is the channel sum of dx)."""

    @staticmethod
    def forward(ctx, x, xb, weight, bias, eps, defer):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_bias_ln_bf16_forward(x2.data_ptr(), _lib.ptr(xb), 1 if (xb is not None and xb.dtype == torch.bfloat16) else 0,
                                                         weight.data_ptr(), bias.data_ptr(), eps, rows, C, y.data_ptr(), mean.data_ptr(),
                                                         rstd.data_ptr(), _lib.current_stream()), "combo_bias_ln_bf16_forward")
        ctx.save_for_backward(x2, xb, mean, rstd, weight)
        ctx.defer, ctx.shape = defer, x.shape
        return y.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, xb, mean, rstd, weight = ctx.saved_tensors
        rows, C = x2.shape
        dy = dy.reshape(rows, C)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        want_param = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dy32 = torch.empty(rows, C, device=x2.device, dtype=torch.float32) if want_param else None
        z32 = torch.empty(rows, C, device=x2.device, dtype=torch.float32) if want_param else None
        dx = torch.empty_like(x2)
        _lib.check(_lib.lib().combo_bias_ln_bf16_backward(dy.data_ptr(), x2.data_ptr(), _lib.ptr(xb),
                                                          1 if (xb is not None and xb.dtype == torch.bfloat16) else 0, mean.data_ptr(),
                                                          rstd.data_ptr(), weight.data_ptr(), rows, C, dx.data_ptr(), _lib.ptr(dy32),
                                                          _lib.ptr(z32), _lib.current_stream()), "combo_bias_ln_bf16_backward")
        dxb = None
        if xb is not None and ctx.needs_input_grad[1]:
            from .colsum import channel_sum
            dxb = channel_sum(dx, rows, C, 1, out_dtype=xb.dtype)
        dw = db = None
        if want_param:
            use = (dy32, z32, mean, rstd)  # (the deferred kernel normalises (z - mean) * rstd itself)
            q = _linear_mod._ln_queue
            out = torch.empty(2, C, device=x2.device, dtype=torch.float32)
            if ctx.defer and q is not None:
                q.append([[use], out])
            else:
                _linear_mod._flush_ln([[[use], out]])
            dw, db = out[0], out[1]
        return dx.view(ctx.shape), dxb, dw, db, None, None


def bias_ln_usable(x, C):
    return x.is_cuda and x.dtype == torch.bfloat16 and C in WIDTHS and C % 4 == 0


def bias_ln(x, xb, norm, defer=True):
    return _BiasLn.apply(x, xb, norm.weight, norm.bias, norm.eps, defer)
