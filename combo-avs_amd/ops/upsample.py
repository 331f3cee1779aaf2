"""Bilinear 2x upsampling (align_corners=False) of channels_last fp32 maps, csrc/upsample.hip (gather-form backward)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


class _Up2(Function):
    """y = up2(x) (+ add): `add` is the lateral branch of the FPN step (msdeformattn.py:350), summed in the same pass; its
    gradient is dy itself."""

    @staticmethod
    def forward(ctx, x, add=None):
        B, C, H, W = x.shape
        y = torch.empty((B, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        _lib.check(_lib.lib().combo_upsample2x_bilinear_add_nhwc_f32(x.data_ptr(), x.stride(0), _lib.ptr(add), B, H, W, C, y.data_ptr(),
                                                                     _lib.current_stream()), "combo_upsample2x_bilinear_add_nhwc_f32")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
            _lib.check(_lib.lib().combo_upsample2x_bilinear_nhwc_backward_f32(dy.data_ptr(), B, H, W, C, dx.data_ptr(), dx.stride(0),
                                                                              _lib.current_stream()),
                       "combo_upsample2x_bilinear_nhwc_backward_f32")
        return dx, (dy if ctx.needs_input_grad[1] else None)


def _token_major(x):
    """[B,C,H,W] view whose memory is [B][H][W][C] with an arbitrary batch stride (channels_last, or a per-level row block of
    the pixel decoder's encoder memory [B, S, C])"""
    B, C, H, W = x.shape
    return x.stride(1) == 1 and x.stride(3) == C and x.stride(2) == W * C and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0


def _up2_ok(x, size):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and tuple(size) == (2 * x.shape[2], 2 * x.shape[3])
            and x.shape[1] % 4 == 0 and _token_major(x))


def upsample_bilinear(x, size):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False); the exact-2x channels_last fp32 case runs on the HIP
    kernels, anything else on ATen."""
    if _up2_ok(x, size):
        return _Up2.apply(x, None)
    return torch.nn.functional.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


FUSE_ADD = True  # the FPN step's lateral add inside the upsampling pass (A/B: tools/ab_const.py)


def upsample_bilinear_add(lateral, x):
    """lateral + F.interpolate(x, size=lateral.shape[-2:], mode="bilinear", align_corners=False) (msdeformattn.py:350): one pass
    when the lateral map is channels_last fp32 and the scale is exactly 2, else the two separate ops."""
    if (FUSE_ADD and _up2_ok(x, lateral.shape[-2:]) and lateral.dtype == torch.float32 and lateral.shape[:2] == x.shape[:2]
            and lateral.is_contiguous(memory_format=torch.channels_last) and lateral.data_ptr() % 16 == 0):
        return _Up2.apply(x, lateral)
    return lateral + upsample_bilinear(x, lateral.shape[-2:])
