"""Depth-wise 3x3 convolution of the PVTv2 MLP on token-major bf16 activations (csrc/dwconv.hip): forward, backward-data
and weight/bias gradient as bandwidth-bound HIP kernels (MIOpen falls back to naive kernels for this op on gfx950)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def usable(x, weight):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous() and x.shape[-1] % 8 == 0
            and weight.dtype == torch.float32 and tuple(weight.shape[1:]) == (1, 3, 3))


class _DWConv3x3(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        """x [B,H,W,C] bf16 contiguous (token-major), weight [C,1,3,3] fp32, bias [C] fp32 -> [B,H,W,C] bf16"""
        B, H, W, C = x.shape
        # tap-major [9][C]: a thread's 8 channels of one tap are 32 contiguous bytes and a wave reads 2 KB in a row (the
        # parameter's own [C][9] layout was measured: every lane then walks its own 288-byte chunk, +2.6 ms per PVTv2 step)
        w = weight.view(C, 9).t().contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.lib().combo_dwconv3x3_bf16(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), B, H, W, C, 0, y.data_ptr(),
                                                   _lib.current_stream()), "combo_dwconv3x3_bf16")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, H, W, C = x.shape
        lib, st = _lib.lib(), _lib.current_stream()
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(lib.combo_dwconv3x3_bf16(dy.data_ptr(), w.data_ptr(), 0, B, H, W, C, 1, dx.data_ptr(), st),
                       "combo_dwconv3x3_bf16")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            slices = lib.combo_dwconv3x3_wgrad_slices(B, H, W, C)
            part = torch.empty(slices, 10, C, device=x.device, dtype=torch.float32)
            _lib.check(lib.combo_dwconv3x3_wgrad_bf16(x.data_ptr(), dy.data_ptr(), B, H, W, C, slices, part.data_ptr(), st),
                       "combo_dwconv3x3_wgrad_bf16")
            dw = torch.empty(C, 1, 3, 3, device=x.device, dtype=torch.float32)
            db = torch.empty(C, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            # one launch: sums the per-slice partials and writes the parameter's layout (v1: two reduce launches + a transpose)
            _lib.check(lib.combo_dwconv3x3_wgrad_finish_f32(part.data_ptr(), slices, C, dw.data_ptr(), _lib.ptr(db), st),
                       "combo_dwconv3x3_wgrad_finish_f32")
        return dx, dw, db


def dwconv3x3(x, weight, bias):
    return _DWConv3x3.apply(x, weight, bias)
