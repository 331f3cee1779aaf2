"""nn.LayerNorm whose PARAMETER gradients can be deferred into one grouped launch (csrc/lngrad.hip) at the end of the
backward pass (ops.linear.deferred_dw).  Forward and the input gradient are ATen's kernels; only the two per-layer
gamma/beta reduction kernels (off the critical path, ~100 launches per step) are replaced."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import linear as _linear_mod


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[-1]
        y, mean, rstd = torch.native_layer_norm(x, (C,), weight, bias, eps)
        ctx.save_for_backward(x, mean, rstd, weight, bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, mean, rstd, weight, bias = ctx.saved_tensors
        C = x.shape[-1]
        dy = dy.contiguous()
        q = _linear_mod._ln_queue
        deferred = (q is not None and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
                    and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]))
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1] and not deferred, ctx.needs_input_grad[2] and not deferred]
        dx, dw, db = torch.ops.aten.native_layer_norm_backward(dy, x, [C], mean, rstd, weight, bias, mask)
        if deferred:
            use = (dy.view(-1, C), x.view(-1, C), mean.reshape(-1), rstd.reshape(-1))
            key = ("ln", weight.data_ptr())
            ent = _linear_mod._dw_index.get(key)
            if ent is not None:  # another application of the same LayerNorm: joins the entry, autograd gets "no gradient"
                ent[0].append(use)
                return dx, None, None, None
            out = torch.empty(2, C, device=x.device, dtype=torch.float32)  # filled when deferred_dw() closes
            ent = [[use], out]
            q.append(ent)
            _linear_mod._dw_index[key] = ent
            dw, db = out[0], out[1]
        return dx, dw, db, None


class LayerNorm(nn.LayerNorm):
    """Same parameters / state-dict as nn.LayerNorm.  `defer_dw = True`: the parameter gradients may join the grouped launch
    of ops.linear.deferred_dw (repeated applications of one instance are summed there)."""
    defer_dw = False

    def forward(self, x):
        if self.defer_dw and x.is_cuda and x.dtype == torch.float32 and self.elementwise_affine and torch.is_grad_enabled() \
                and not torch.is_autocast_enabled():
            return _LayerNorm.apply(x.contiguous(), self.weight, self.bias, self.eps)
        return super().forward(x)
