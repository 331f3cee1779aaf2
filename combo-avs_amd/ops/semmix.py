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
                                        _lib.ptr(o1), _lib.ptr(o2), _lib.current_stream()), "combo_sem_mix")


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


class _GateMix(Function):
    """out = f + gate(mean_HW(p)) * p as ONE autograd node (models/utils/misc.py:112-131 + maskformer_model.py:345-352).  As
    separate nodes (pool, gate, mix) the gradient of p arrived in two pieces - dout * s from the mix and the pool's broadcast
    gradient - which autograd added with a full-map kernel per level; here the pool's gradient [B, C] rides in the mix-backward
    kernel (its `dgap` operand) and, in fp32, df IS dout (no copy).  `gate` is a callable [B, C] -> [B, C] built from ordinary
    autograd ops (the two small dense layers + sigmoid): its graph is recorded inside forward and differentiated inside
    backward; gate_params are its parameters (listed so that autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, f, p, gate, record, *gate_params):
        f, B, HW, C = _cl(f)
        p, _, _, _ = _cl(p)
        _lib.require_cuda(f, p, channels_last=True)
        acc = torch.empty(B, C, device=p.device, dtype=torch.float32)
        _call(0, p.dtype == torch.bfloat16, p, None, None, None, B, HW, C, acc)
        # record (decided by gate_mix, where the caller's grad mode is still visible): False under no_grad / inference_mode
        # (evaluation, bench --mode infer) - then no inner graph is built or kept
        if record:
            with torch.enable_grad():
                gap = (acc / HW).requires_grad_(True)
                s = gate(gap)
        else:
            with torch.no_grad():
                gap, s = None, gate(acc / HW)
        s_val = s.detach().contiguous().float()
        out = torch.empty(f.shape, device=f.device, dtype=torch.float32, memory_format=torch.channels_last)
        _call(1, f.dtype == torch.bfloat16, f, p, s_val, None, B, HW, C, out)
        if record:
            ctx.save_for_backward(p, s_val, *gate_params)  # the parameters travel through save_for_backward (version checks, hooks)
            ctx.inner = (gap, s)
        ctx.f_dtype = f.dtype
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        p, s_val, *params = ctx.saved_tensors
        gap, s = ctx.inner
        ctx.inner = None
        dout, B, HW, C = _cl(dout.float())
        is_bf16 = p.dtype == torch.bfloat16
        ds = torch.empty(B, C, device=p.device, dtype=torch.float32)
        _call(2, is_bf16, p, dout, None, None, B, HW, C, ds)
        need = [t for t in params if t.requires_grad]
        with torch.enable_grad():
            grads = torch.autograd.grad(s, [gap] + need, ds.to(s.dtype))
        dgap = (grads[0].float() / HW).contiguous()
        it = iter(grads[1:])
        dparams = [next(it) if t.requires_grad else None for t in params]
        dp = torch.empty_like(p)
        if ctx.f_dtype == torch.float32 and not is_bf16:
            df = dout  # d out / d f = 1
            _call(3, False, dout, None, s_val, dgap, B, HW, C, None, dp)
        else:
            df = torch.empty_like(p)
            _call(3, is_bf16, dout, None, s_val, dgap, B, HW, C, df, dp)
        return (df, dp, None, None) + tuple(dparams)


def gate_mix(f, p, gate, gate_params):
    record = torch.is_grad_enabled() and any(t.requires_grad for t in (f, p, *gate_params))
    return _GateMix.apply(f, p, gate, record, *gate_params)


def global_avg_pool(p):
    return _Gap.apply(p)


def mix(f, p, s):
    return _Mix.apply(f, p, s)
