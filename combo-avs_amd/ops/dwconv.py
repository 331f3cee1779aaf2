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
        wT = weight.view(C, 9).t().contiguous()  # tap-major [9][C]
        y = torch.empty_like(x)
        _lib.check(_lib.lib().combo_dwconv3x3_bf16(x.data_ptr(), wT.data_ptr(), _lib.ptr(bias), B, H, W, C, 0, y.data_ptr(),
                                                   _lib.current_stream()), "combo_dwconv3x3_bf16")
        ctx.save_for_backward(x, wT)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wT = ctx.saved_tensors
        B, H, W, C = x.shape
        lib, st = _lib.lib(), _lib.current_stream()
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(lib.combo_dwconv3x3_bf16(dy.data_ptr(), wT.data_ptr(), 0, B, H, W, C, 1, dx.data_ptr(), st),
                       "combo_dwconv3x3_bf16")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            slices = lib.combo_dwconv3x3_wgrad_slices(B, H, W, C)
            # two-level sum of the per-slice partials: slice s = s2 * 16 + s1; pass 1 sums over s2 (many threads), pass 2
            # over the 16 s1 (the split-K reduce kernel walks its split axis serially, so that axis must stay short)
            slices = -(-slices // 16) * 16
            part = torch.empty(slices, 10, C, device=x.device, dtype=torch.float32)
            _lib.check(lib.combo_dwconv3x3_wgrad_bf16(x.data_ptr(), dy.data_ptr(), B, H, W, C, slices, part.data_ptr(), st),
                       "combo_dwconv3x3_wgrad_bf16")
            tmp = torch.empty(16, 10, C, device=x.device, dtype=torch.float32)
            tot = torch.empty(10, C, device=x.device, dtype=torch.float32)
            _lib.check(lib.combo_splitk_reduce_f32(part.data_ptr(), slices // 16, 16 * 10 * C, tmp.data_ptr(), 0, 0, 0, st),
                       "combo_splitk_reduce_f32")
            _lib.check(lib.combo_splitk_reduce_f32(tmp.data_ptr(), 16, 10 * C, tot.data_ptr(), 0, 0, 0, st),
                       "combo_splitk_reduce_f32")
            dw = tot[:9].t().reshape(C, 1, 3, 3)
            db = tot[9] if ctx.has_bias else None
        return dx, dw, db


def dwconv3x3(x, weight, bias):
    return _DWConv3x3.apply(x, weight, bias)
