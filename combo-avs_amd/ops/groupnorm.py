"""GroupNorm (+ ReLU) on channels_last fp32 maps (csrc/groupnorm.hip): no NCHW round trip, channels_last in and out."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def usable(x, gn):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and x.shape[1] % 4 == 0 and x.shape[1] <= 1024 and gn.affine and not torch.is_autocast_enabled())


class _GroupNormNHWC(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu):
        B, C, H, W = x.shape
        lib = _lib.lib()
        slices = lib.combo_groupnorm_nhwc_slices(H * W)
        dev = x.device
        y = torch.empty_like(x)  # preserves channels_last
        mean = torch.empty(B, groups, device=dev, dtype=torch.float32)
        rstd = torch.empty(B, groups, device=dev, dtype=torch.float32)
        part = torch.empty(B * slices * C * 2, device=dev, dtype=torch.float32)
        _lib.check(lib.combo_groupnorm_nhwc_forward_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), B, H * W, C, groups, eps,
                                                        1 if relu else 0, part.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                        y.data_ptr(), _lib.current_stream()), "combo_groupnorm_nhwc_forward_f32")
        ctx.save_for_backward(x, y if relu else None, mean, rstd, weight)
        ctx.groups, ctx.relu = groups, relu
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, mean, rstd, weight = ctx.saved_tensors
        B, C, H, W = x.shape
        lib = _lib.lib()
        dev = x.device
        dy = dy.contiguous(memory_format=torch.channels_last)
        slices = lib.combo_groupnorm_nhwc_slices(H * W)
        part = torch.empty(B * slices * C * 2, device=dev, dtype=torch.float32)
        s12 = torch.empty(B * ctx.groups * 2 + B * C * 2, device=dev, dtype=torch.float32)
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, device=dev, dtype=torch.float32)
        dbeta = torch.empty(C, device=dev, dtype=torch.float32)
        _lib.check(lib.combo_groupnorm_nhwc_backward_f32(dy.data_ptr(), x.data_ptr(), _lib.ptr(y), mean.data_ptr(), rstd.data_ptr(),
                                                         weight.data_ptr(), B, H * W, C, ctx.groups, 1 if ctx.relu else 0,
                                                         part.data_ptr(), s12.data_ptr(), dx.data_ptr(), dgamma.data_ptr(),
                                                         dbeta.data_ptr(), _lib.current_stream()), "combo_groupnorm_nhwc_backward_f32")
        return dx, dgamma, dbeta, None, None, None


def group_norm(x, gn, relu=False):
    """x: channels_last fp32 [B,C,H,W]; gn: nn.GroupNorm -> channels_last output (ReLU fused when asked)."""
    return _GroupNormNHWC.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)
