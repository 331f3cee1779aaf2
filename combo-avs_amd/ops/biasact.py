"""Fused epilogue of the backbone convolutions (csrc/biasact.hip): y <- relu(y + bias[c] (+ residual)), in place on the
convolution output (bf16 or fp32, channels_last), one pass; backward dx = dy * (y > 0) for both the convolution branch and the
residual branch."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def fusable(y, residual=None):
    ok = (y.is_cuda and y.dtype in (torch.bfloat16, torch.float32) and y.dim() == 4 and y.shape[1] % 8 == 0
          and y.is_contiguous(memory_format=torch.channels_last))
    if residual is not None:
        ok = ok and residual.dtype == y.dtype and residual.shape == y.shape and residual.is_contiguous(memory_format=torch.channels_last)
    return ok


class _BiasAct(Function):
    @staticmethod
    def forward(ctx, y, bias, residual, relu, fanout=False, grad_masked=False, precomputed=False):
        """precomputed: y already IS relu(conv + bias (+ residual)) - the producing GEMM applied the epilogue (ops/convwrw.py) -
        and this node only routes the gradients (ReLU mask, residual branch, fan-out sum)"""
        N, C, H, W = y.shape
        if not precomputed:
            fn = _lib.lib().combo_bias_act_f32 if y.dtype == torch.float32 else _lib.lib().combo_bias_act_bf16
            _lib.check(fn(y.data_ptr(), bias.float().data_ptr() if bias.dtype != torch.float32 else bias.data_ptr(), _lib.ptr(residual),
                          N * H * W, C, 1 if relu else 0, _lib.current_stream()), "combo_bias_act")
            ctx.mark_dirty(y)
        # grad_masked: the single consumer hands back a gradient that is already multiplied by [y > 0] (ops/convwrw.py)
        ctx.relu, ctx.has_res, ctx.fanout = relu and not grad_masked, residual is not None, fanout
        if ctx.relu:
            ctx.save_for_backward(y)
        if fanout:
            # the SAME values as two autograd outputs (the second is a view): a consumer pair - next block's first convolution and
            # its identity / shortcut branch - then hands back two separate gradients, added inside the ReLU-gradient pass
            return y, y.view_as(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dy2=None):
        if dy is None:
            dy, dy2 = dy2, None
        dx = dy
        if ctx.relu:
            (y,) = ctx.saved_tensors
            dy = dy.contiguous(memory_format=torch.channels_last)
            dx = torch.empty_like(dy)
            if dy2 is not None and dy.dtype == torch.float32 and dy2.dtype == torch.float32:
                dy2 = dy2.contiguous(memory_format=torch.channels_last)
                _lib.check(_lib.lib().combo_relu_grad2_f32(dy.data_ptr(), dy2.data_ptr(), y.data_ptr(), dy.numel(), dx.data_ptr(),
                                                           _lib.current_stream()), "combo_relu_grad2_f32")
            else:
                if dy2 is not None:
                    dy = (dy + dy2).contiguous(memory_format=torch.channels_last)
                fn = _lib.lib().combo_relu_grad_f32 if dy.dtype == torch.float32 else _lib.lib().combo_relu_grad_bf16
                _lib.check(fn(dy.data_ptr(), y.data_ptr(), dy.numel(), dx.data_ptr(), _lib.current_stream()), "combo_relu_grad")
        elif dy2 is not None:
            dx = dy + dy2
        return dx, None, (dx if ctx.has_res else None), None, None, None, None


def bias_act(y, bias, residual=None, relu=True, fanout=False, grad_masked=False, precomputed=False):
    """y: convolution output WITHOUT bias (modified in place); bias: fp32 [C]; residual: same shape as y or None.
    fanout: return the result twice (two autograd outputs over the same memory) for a consumer pair, see _BiasAct.forward."""
    if fusable(y, residual):
        return _BiasAct.apply(y, bias, residual, relu, fanout, grad_masked, precomputed)
    assert not precomputed
    out = y + bias.to(y.dtype)[None, :, None, None]
    if residual is not None:
        out = out + residual
    out = torch.relu_(out) if relu else out
    return (out, out) if fanout else out
