"""Siam-Encoder-Module mix on channels-last activations (csrc/semmix.hip)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def _cl(x):
    """[B,C,H,W] -> channels-last memory, returns (tensor, B, HW, C)"""
    x = x.contiguous(memory_format=torch.channels_last)
    B, C, H, W = x.shape
    return x, B, H * W, C


def _call(op, is_bf16, a, b, s, g, B, HW, C, o1, o2=None):
    _lib.check(_lib.lib().combo_sem_mix(op, 1 if is_bf16 else 0, a.data_ptr(), _lib.ptr(b), _lib.ptr(s), _lib.ptr(g), B, HW, C,
                                        o1.data_ptr(), _lib.ptr(o2), _lib.current_stream()), "combo_sem_mix")


class _Gap(Function):
    """mean over H,W of a channels-last [B,C,H,W] tensor -> [B,C] fp32"""

    @staticmethod
    def forward(ctx, p):
        p, B, HW, C = _cl(p)
        _lib.require_cuda(p, channels_last=True)
        acc = torch.empty(B, C, device=p.device, dtype=torch.float32)  # (written, not accumulated: csrc/semmix.hip sem_reduce)
        _call(0, p.dtype == torch.bfloat16, p, None, None, None, B, HW, C, acc)
        ctx.shape, ctx.dtype = p.shape, p.dtype
        return acc / HW

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        return (g / (H * W)).to(ctx.dtype)[:, :, None, None].expand(B, C, H, W)


class _Mix(Function):
    """out = f + s[:, :, None, None] * p  (fp32, channels-last)"""

    @staticmethod
    def forward(ctx, f, p, s):
        f, B, HW, C = _cl(f)
        p, _, _, _ = _cl(p)
        s = s.contiguous().float()
        _lib.require_cuda(f, p, channels_last=True)
        _lib.require_cuda(s)
        out = torch.empty(f.shape, device=f.device, dtype=torch.float32, memory_format=torch.channels_last)
        _call(1, f.dtype == torch.bfloat16, f, p, s, None, B, HW, C, out)
        ctx.save_for_backward(p, s)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        p, s = ctx.saved_tensors
        dout, B, HW, C = _cl(dout.float())
        is_bf16 = p.dtype == torch.bfloat16
        ds = torch.empty(B, C, device=p.device, dtype=torch.float32)
        _call(2, is_bf16, p, dout, None, None, B, HW, C, ds)
        df = torch.empty_like(p)
        dp = torch.empty_like(p)
        zero = torch.zeros(B, C, device=p.device, dtype=torch.float32)
        _call(3, is_bf16, dout, None, s, zero, B, HW, C, df, dp)
        return df, dp, ds


def global_avg_pool(p):
    return _Gap.apply(p)


def mix(f, p, s):
    return _Mix.apply(f, p, s)
