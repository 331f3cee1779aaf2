"""Small building blocks shared by the head modules."""
import math

import torch
from torch import nn
from torch.nn import functional as F

_PE_CACHE = {}


def position_embedding_sine(b, h, w, device, num_pos_feats=128, temperature=10000.0, dtype=torch.float32):
    """PositionEmbeddingSine(normalize=True) of the reference (position_encoding.py:29-48).
    Input independent, so it is computed once per (h, w, device) and cached (the reference recomputes it
    7 times per forward).  Returns [1, 2*num_pos_feats, h, w]; callers broadcast over the batch."""
    key = (h, w, str(device), num_pos_feats)
    pe = _PE_CACHE.get(key)
    if pe is None:
        eps, scale = 1e-6, 2 * math.pi
        y = torch.arange(1, h + 1, dtype=torch.float32, device=device).view(h, 1).expand(h, w)
        x = torch.arange(1, w + 1, dtype=torch.float32, device=device).view(1, w).expand(h, w)
        y = y / (float(h) + eps) * scale
        x = x / (float(w) + eps) * scale
        i = torch.arange(num_pos_feats, dtype=torch.float32, device=device)
        dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
        px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
        px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
        py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
        pe = torch.cat((py, px), dim=2).permute(2, 0, 1).unsqueeze(0).contiguous()
        _PE_CACHE[key] = pe
    return pe.to(dtype)


def conv1x1_or_conv(conv, x):
    """A stride-1 1x1 convolution on a channels_last fp32 tensor IS a token-major GEMM [B*H*W, Cin] x [Cin, Cout]: route it
    through ops.linear (csrc/gemm_nt.hip forward / dX, deferred grouped weight gradient) - measured 102 + 220 us against
    MIOpen's fp32 245 + 401 us at 40 x 256 x 56 x 56 (tools/bench_conv1x1.py).  The 3x3 / stride 1 / pad 1 FPN output
    convolution runs as an implicit GEMM on the same kernels (ops/conv3x3.py).  Everything else goes to MIOpen."""
    if (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and x.is_cuda
            and x.dtype == torch.float32 and not torch.is_autocast_enabled() and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 16 == 0):
        from ..ops.linear import linear
        B, C, H, W = x.shape
        y = linear(x.permute(0, 2, 3, 1).reshape(B * H * W, C), conv.weight.view(conv.out_channels, C), conv.bias, defer=True)
        return y.view(B, H, W, conv.out_channels).permute(0, 3, 1, 2)  # NCHW view, channels_last memory
    if conv.kernel_size == (3, 3):
        from ..ops import conv3x3
        if conv3x3.ENABLED and conv3x3.usable(conv, x):
            return conv3x3.conv3x3(x, conv.weight, conv.bias)  # implicit GEMM on csrc/gemm_nt.hip / gemm_tn.hip
    if x.is_cuda:
        from .. import _lib
        _lib.fallback_notice(f"modeling.layers.conv1x1_or_conv[w {tuple(conv.weight.shape)}]",
                             f"x {x.dtype} {tuple(x.shape)}, channels_last {x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)}, "
                             f"autocast {torch.is_autocast_enabled()}: MIOpen convolution")
    return F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)


def norm_act(x, norm, activation=None):
    """norm -> activation of detectron2's Conv2d wrapper; GroupNorm (+ ReLU) on a channels_last fp32 map is one fused
    channels_last-in / channels_last-out op (csrc/groupnorm.hip), everything else goes through the modules."""
    if isinstance(norm, nn.GroupNorm):
        from ..ops import groupnorm
        if groupnorm.usable(x, norm) and activation in (None, F.relu):
            return groupnorm.group_norm(x, norm, relu=activation is F.relu)
    if norm is not None:
        x = norm(x)
    if activation is not None:
        x = activation(x)
    return x


class Conv2d(nn.Conv2d):
    """nn.Conv2d + optional norm + optional activation; parameters are named like detectron2's wrapper
    (`<name>.weight`, `<name>.norm.weight`) so reference checkpoints load 1:1."""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = conv1x1_or_conv(self, x)
        return norm_act(x, self.norm, self.activation)


def get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    raise ValueError(f"unsupported norm {norm!r}")


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class MLP(nn.Module):
    """transformer_decoder.py:207-219"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        from ..ops.linear import linear
        for i, layer in enumerate(self.layers):
            # defer=True: the weight gradients may join ops.linear.deferred_dw's grouped launch (repeated uses are summed there)
            x = linear(x, layer.weight, layer.bias, relu=i < self.num_layers - 1, defer=True)
        return x
