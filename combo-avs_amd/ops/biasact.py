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
            # the SAME values as two (fanout = True / 2) or three autograd outputs (views): the consumers - next block's first
            # convolution, its identity / shortcut branch, and for a stage's last block the head - then hand back separate
            # gradients, added inside the ReLU-gradient pass
            return (y,) + tuple(y.view_as(y) for _ in range(max(int(fanout), 2) - 1))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        got = [g for g in dys if g is not None]
        dy = got[0]
        dx = dy
        if ctx.relu:
            (y,) = ctx.saved_tensors
            f32 = all(g.dtype == torch.float32 for g in got)
            got = [g.contiguous(memory_format=torch.channels_last) for g in got]
            dy = got[0]
            dx = torch.empty_like(dy)
            lib, st = _lib.lib(), _lib.current_stream()
            if len(got) == 3 and f32:
                _lib.check(lib.combo_relu_grad3_f32(got[0].data_ptr(), got[1].data_ptr(), got[2].data_ptr(), y.data_ptr(), dy.numel(),
                                                    dx.data_ptr(), st), "combo_relu_grad3_f32")
            elif len(got) == 2 and f32:
                _lib.check(lib.combo_relu_grad2_f32(got[0].data_ptr(), got[1].data_ptr(), y.data_ptr(), dy.numel(), dx.data_ptr(), st),
                           "combo_relu_grad2_f32")
            else:
                for g in got[1:]:
                    dy = (dy + g).contiguous(memory_format=torch.channels_last)
                fn = lib.combo_relu_grad_f32 if dy.dtype == torch.float32 else lib.combo_relu_grad_bf16
                _lib.check(fn(dy.data_ptr(), y.data_ptr(), dy.numel(), dx.data_ptr(), st), "combo_relu_grad")
        else:
            for g in got[1:]:
                dx = dx + g
        return dx, None, (dx if ctx.has_res else None), None, None, None, None


class _MaskedFan(Function):
    """x (a ReLU output whose producer was told grad_masked=True) -> two aliases for its two consumers; the gradients come back
    summed and multiplied by [x > 0] in one pass - the producer-side fan-out of _BiasAct moved to the consumer, for the layers
    ops.convwrw._ConvWrw's `passthrough` form does not take."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, d1, d2):
        (x,) = ctx.saved_tensors
        got = [g.contiguous(memory_format=torch.channels_last) for g in (d1, d2) if g is not None]
        if not got:
            return None
        lib, st = _lib.lib(), _lib.current_stream()
        dx = torch.empty_like(got[0])
        if len(got) == 2 and all(g.dtype == torch.float32 for g in got) and x.dtype == torch.float32:
            _lib.check(lib.combo_relu_grad2_f32(got[0].data_ptr(), got[1].data_ptr(), x.data_ptr(), dx.numel(), dx.data_ptr(), st),
                       "combo_relu_grad2_f32")
            return dx
        dy = got[0] if len(got) == 1 else (got[0] + got[1]).contiguous(memory_format=torch.channels_last)
        fn = lib.combo_relu_grad_f32 if dy.dtype == torch.float32 else lib.combo_relu_grad_bf16
        _lib.check(fn(dy.data_ptr(), x.data_ptr(), dy.numel(), dx.data_ptr(), st), "combo_relu_grad")
        return dx


def masked_fan(x):
    return _MaskedFan.apply(x)


def bias_act(y, bias, residual=None, relu=True, fanout=False, grad_masked=False, precomputed=False):
    """y: convolution output WITHOUT bias (modified in place); bias: fp32 [C]; residual: same shape as y or None.
    fanout: return the result twice (True / 2) or three times (two / three autograd outputs over the same memory) for its
    consumers, see _BiasAct.forward."""
    if fusable(y, residual):
        return _BiasAct.apply(y, bias, residual, relu, fanout, grad_masked, precomputed)
    assert not precomputed
    out = y + bias.to(y.dtype)[None, :, None, None]
    if residual is not None:
        out = out + residual
    out = torch.relu_(out) if relu else out
    return (out,) * max(int(fanout), 2) if fanout else out
