"""Bilinear 2x upsampling (align_corners=False) of channels_last fp32 maps, csrc/upsample.hip (gather-form backward)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


class _Up2(Function):
    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        y = torch.empty((B, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        _lib.check(_lib.lib().combo_upsample2x_bilinear_nhwc_f32(x.data_ptr(), x.stride(0), B, H, W, C, y.data_ptr(),
                                                                 _lib.current_stream()), "combo_upsample2x_bilinear_nhwc_f32")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
        _lib.check(_lib.lib().combo_upsample2x_bilinear_nhwc_backward_f32(dy.data_ptr(), B, H, W, C, dx.data_ptr(), dx.stride(0),
                                                                          _lib.current_stream()),
                   "combo_upsample2x_bilinear_nhwc_backward_f32")
        return dx


def _token_major(x):
    """[B,C,H,W] view whose memory is [B][H][W][C] with an arbitrary batch stride (channels_last, or a per-level row block of
    the pixel decoder's encoder memory [B, S, C])"""
    B, C, H, W = x.shape
    return x.stride(1) == 1 and x.stride(3) == C and x.stride(2) == W * C and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0


def upsample_bilinear(x, size):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False); the exact-2x channels_last fp32 case runs on the HIP
    kernels, anything else on ATen."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and tuple(size) == (2 * x.shape[2], 2 * x.shape[3])
            and x.shape[1] % 4 == 0 and _token_major(x)):
        return _Up2.apply(x)
    return torch.nn.functional.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)
