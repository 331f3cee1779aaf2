"""Residual add + LayerNorm of the head's post-norm layers on csrc/layernorm.hip (one pass forward, one pass backward), with
the PARAMETER gradients deferred into one grouped launch (csrc/lngrad.hip) at the end of the backward pass
(ops.linear.deferred_dw).  `LayerNorm(x, residual=r)` == nn.LayerNorm(x + r)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from . import linear as _linear_mod


def _own_ok(x, C):
    return (x.is_cuda and x.dtype == torch.float32 and C in (64, 128, 256, 320, 512) and not torch.is_autocast_enabled())


class _AddLayerNorm(Function):
    """fanout > 1: the result is returned `fanout` times (aliases of one buffer, one autograd output per consumer) so that the
    consumers' gradients arrive separately and are summed INSIDE the backward kernel instead of by autograd's accumulation
    kernels; pos [rows_per_frame, C]: one more output, y + pos (the next block's query input)."""

    @staticmethod
    def forward(ctx, x, r, weight, bias, eps, defer, fanout=1, pos=None):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        rows = x2.shape[0]
        r2 = None
        if r is not None:
            r2 = r.reshape(-1, C)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        y = torch.empty_like(x2)
        z = torch.empty_like(x2) if r is not None else None  # the normalised tensor the backward pass needs (z = x without r)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        yp = pos2 = None
        if pos is not None:
            pos2 = pos.reshape(-1, C)
            pos2 = pos2 if pos2.is_contiguous() else pos2.contiguous()
            assert rows % pos2.shape[0] == 0, "pos is broadcast over whole frames"
            yp = torch.empty_like(x2)
        _lib.check(_lib.lib().combo_add_layernorm_forward_f32(x2.data_ptr(), _lib.ptr(r2), weight.data_ptr(), bias.data_ptr(), eps, rows, C,
                                                              _lib.ptr(z), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                              _lib.ptr(pos2), pos2.shape[0] if pos2 is not None else 0, _lib.ptr(yp),
                                                              _lib.current_stream()), "combo_add_layernorm_forward_f32")
        ctx.save_for_backward(z if z is not None else x2, mean, rstd, weight)
        ctx.has_r, ctx.defer, ctx.shape = r is not None, defer, x.shape
        ctx.n_out = fanout + (1 if pos is not None else 0)
        ctx.pos_rows = pos2.shape[0] if pos2 is not None else 0
        ctx.pos_shape = pos.shape if pos is not None else None
        if ctx.n_out == 1:
            return y.view(x.shape)
        y = y.view(x.shape)
        outs = [y] + [y.view_as(y) for _ in range(fanout - 1)]
        if yp is not None:
            outs.append(yp.view(x.shape))
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        z, mean, rstd, weight = ctx.saved_tensors
        C = z.shape[1]
        gs = []
        for d in dys:  # the gradients of the aliases (and of y + pos: d(y + pos)/dy = 1); unused outputs arrive as None
            if d is not None:
                d = d.reshape(-1, C)
                gs.append(d if d.is_contiguous() else d.contiguous())
        if not gs:
            return (None,) * 8
        dpos = None
        if ctx.pos_rows and ctx.needs_input_grad[7] and dys[-1] is not None:
            # d(y + pos)/dpos = 1, broadcast over the frames: dpos = the y + pos output's gradient summed over the frames
            # (csrc/colsum.hip: no memset node in a captured step); pos is learnable at every call site (query_embed, the
            # level embedding inside the pixel decoder's position rows)
            from .colsum import channel_sum
            dyp = gs[-1]
            dpos = channel_sum(dyp, dyp.shape[0] // ctx.pos_rows, ctx.pos_rows * C, 1).view(ctx.pos_shape)
        while len(gs) > 4:  # (never with this model's fan-outs <= 4)
            gs = [gs[0] + gs[1]] + gs[2:]
        want_param = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dy2 = gs[0]
        dy_sum = torch.empty_like(z) if (len(gs) > 1 and want_param) else None
        dz = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or dy_sum is not None:
            dz = torch.empty_like(z)
            extra = [_lib.ptr(gs[i]) if i < len(gs) else None for i in (1, 2, 3)]
            _lib.check(_lib.lib().combo_layernorm_backward_f32(dy2.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                               weight.data_ptr(), z.shape[0], C, dz.data_ptr(), extra[0], extra[1],
                                                               extra[2], _lib.ptr(dy_sum), _lib.current_stream()),
                       "combo_layernorm_backward_f32")
            dz = dz.view(ctx.shape)
        elif len(gs) > 1:
            dy_sum = sum(gs[1:], gs[0])
        if dy_sum is not None:
            dy2 = dy_sum
        dw = db = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            q = _linear_mod._ln_queue
            use = (dy2, z, mean, rstd)
            if ctx.defer and q is not None:
                key = ("ln", weight.data_ptr())
                ent = _linear_mod._dw_index.get(key)
                if ent is not None:  # another application of the same LayerNorm: joins the entry, autograd gets "no gradient"
                    ent[0].append(use)
                else:
                    out = torch.empty(2, C, device=z.device, dtype=torch.float32)  # filled when deferred_dw() closes
                    ent = [[use], out]
                    q.append(ent)
                    _linear_mod._dw_index[key] = ent
                    dw, db = out[0], out[1]
            else:
                out = torch.empty(2, C, device=z.device, dtype=torch.float32)
                _linear_mod._flush_ln([[[use], out]])
                dw, db = out[0], out[1]
        return dz, (dz if ctx.has_r else None), dw, db, None, None, None, dpos


class LayerNorm(nn.LayerNorm):
    """Same parameters / state-dict as nn.LayerNorm; `forward(x, residual=None)` = LN(x + residual).  `defer_dw = True`: the
    parameter gradients may join the grouped launch of ops.linear.deferred_dw (repeated applications of one instance are summed
    there)."""
    defer_dw = False

    def forward(self, x, residual=None, fanout=1, pos=None):
        """fanout / pos: see _AddLayerNorm - returns a tuple of `fanout` aliases of the result (+ result + pos) when fanout > 1
        or pos is given: hand each consumer its own element."""
        if self.elementwise_affine and _own_ok(x, x.shape[-1]) and (residual is None or residual.shape == x.shape):
            return _AddLayerNorm.apply(x, residual, self.weight, self.bias, self.eps, self.defer_dw, fanout, pos)
        y = super().forward(x if residual is None else x + residual)
        if fanout == 1 and pos is None:
            return y
        return tuple([y] * fanout + ([y + pos] if pos is not None else []))
