"""MSDeformAttn pixel decoder, mirroring
models/modeling/pixel_decoder/msdeformattn.py and ops/modules/ms_deform_attn.py of the reference:
same class names, constructor arguments, parameter names and `forward_features` contract.

MI355X notes: the deformable core runs on the hand-written HIP kernels (`combo_avs_amd.msda`), there is no
grid_sample fallback and no bare `except` (ms_deform_attn.py:119-125 of the reference hides kernel failures);
position encodings and reference points are cached per resolution; tokens stay [BT, S, 256] row-major
(one 128-B row per head) which is the layout the LDS-staged kernels consume.
"""
import math
import os
from typing import Dict

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn.init import constant_, normal_, xavier_uniform_

from ..msda import MSDeformAttnFunction
from ..registry import SEM_SEG_HEADS_REGISTRY, ShapeSpec
from ..ops.linear import Linear, ffn, linear
from ..ops.upsample import upsample_bilinear, upsample_bilinear_add  # noqa: F401
from .layers import conv1x1_or_conv, norm_act, Conv2d, c2_xavier_fill, get_norm, position_embedding_sine


# Finer-grained arithmetic switches of the pixel decoder's forward GEMMs / convolutions than the head-wide ops.linear.FORWARD_PRECISION
# (ops.linear.forward_precision_scope: a scope can only move towards the cheaper mode): the deformable transformer encoder's linears, and
# the input projections / FPN lateral + output convolutions / mask-feature projection.  "fp32" = no effect.  tools/probe_flip_luck.py sets
# them to show how the count of flipped attention-mask cells moves with the arithmetic of single groups of layers.
PIXEL_DECODER_FORWARD = "fp32"
PIXEL_DECODER_FPN_FORWARD = "fp32"


def _deferred_layer_norm(dim):
    """per-layer post-norm: applied once per forward, so its parameter gradients may join the grouped launch"""
    from ..ops.layernorm import LayerNorm
    ln = LayerNorm(dim)
    ln.defer_dw = True
    return ln


class MSDeformAttn(nn.Module):
    """ops/modules/ms_deform_attn.py:32-129"""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = Linear(d_model, d_model)
        self.output_proj = Linear(d_model, d_model)
        self.value_proj.defer_dw = self.output_proj.defer_dw = True  # used once per forward: dW may join the grouped launch
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2)
        grid_init = grid_init.repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, offset_normalizer=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        if reference_points.shape[-1] != 2:
            raise ValueError(f"Last dim of reference_points must be 2, but get {reference_points.shape[-1]} instead.")
        if offset_normalizer is None:
            offset_normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
        fused = (query.is_cuda and query.dtype == torch.float32 and not torch.is_autocast_enabled()
                 and self.n_levels * self.n_points <= 16 and self.sampling_offsets.bias is not None)
        if fused:
            # row a5 as two launches: ONE GEMM for both projections + the prologue kernel (loc = ref + off / (W,H); softmax)
            from ..ops.linear import linear_cat
            from ..ops.msdaprep import msda_prep
            proj = linear_cat(query, self.sampling_offsets.weight, self.sampling_offsets.bias,
                              self.attention_weights.weight, self.attention_weights.bias, defer=True)
            sampling_locations, attention_weights = msda_prep(proj, reference_points, offset_normalizer, self.n_heads,
                                                              self.n_levels, self.n_points)
        else:
            sampling_offsets = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
            attention_weights = self.attention_weights(query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)
            attention_weights = F.softmax(attention_weights, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
            sampling_locations = (reference_points[:, :, None, :, None, :]
                                  + sampling_offsets / offset_normalizer[None, None, None, :, None, :])
        output = MSDeformAttnFunction.apply(value.float(), input_spatial_shapes, input_level_start_index,
                                            sampling_locations.float(), attention_weights.float(), self.im2col_step)
        return self.output_proj(output.to(query.dtype))


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    """msdeformattn.py:98-134 (post-norm, dropout = MASK_FORMER.DROPOUT = 0 in every shipped config)"""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = _deferred_layer_norm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.linear2.defer_dw = True
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = _deferred_layer_norm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None, normalizer=None,
                fan=None, last=True):
        """fan = (residual, value input, query = src + pos): three autograd handles on the previous layer's output (aliases of
        one buffer + the `+ pos` variant written by its LayerNorm kernel, ops/layernorm.py) - their gradients are summed inside
        the LayerNorm backward kernel instead of by autograd's accumulation kernels.  last = False: returns such a triple."""
        src_res, src_val, src_q = fan if fan is not None else (src, src, src + pos)
        src2 = self.self_attn(src_q, reference_points, src_val, spatial_shapes, level_start_index, padding_mask, normalizer)
        x_ffn, x_res = self.norm1(src_res, self.dropout1(src2), fanout=2)  # LN(src + src2) in one pass (csrc/layernorm.hip)
        if self.dropout2.p == 0.0:  # (every shipped config) FFN with the ReLU backward folded into linear2's dX GEMM
            src2 = ffn(x_ffn, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)
        else:
            src2 = self.linear2(self.dropout2(linear(x_ffn, self.linear1.weight, self.linear1.bias, relu=True, defer=True)))
        if last:
            return self.norm2(x_res, self.dropout3(src2))
        return self.norm2(x_res, self.dropout3(src2), fanout=2, pos=pos)


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer_fn, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([encoder_layer_fn() for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes_list, device):
        """msdeformattn.py:144-157 with valid_ratios == 1 -> [1, S, L, 2]"""
        pts = []
        for (H, W) in spatial_shapes_list:
            ry = (torch.arange(H, dtype=torch.float32, device=device) + 0.5) / H
            rx = (torch.arange(W, dtype=torch.float32, device=device) + 0.5) / W
            gy, gx = torch.meshgrid(ry, rx, indexing="ij")
            pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
        ref = torch.cat(pts, 0)
        return ref[None, :, None, :].expand(1, -1, len(spatial_shapes_list), -1).contiguous()


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    """msdeformattn.py:23-95"""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = MSDeformAttnTransformerEncoder(
            lambda: MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                        num_feature_levels, nhead, enc_n_points),
            num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._geom = {}
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    def _geometry(self, shapes_list, device):
        key = (tuple(shapes_list), str(device))
        g = self._geom.get(key)
        if g is None:
            spatial_shapes = torch.as_tensor(shapes_list, dtype=torch.long, device=device)
            level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
            ref = MSDeformAttnTransformerEncoder.get_reference_points(shapes_list, device)
            normalizer = torch.tensor([[w, h] for h, w in shapes_list], dtype=torch.float32, device=device)
            g = (spatial_shapes, level_start_index, ref, normalizer)
            self._geom[key] = g
            from .. import msda
            msda.register_level_shapes(spatial_shapes, shapes_list)  # host copy for the fused backward (no sync)
        return g

    def forward(self, srcs, pos_embeds):
        """srcs: list of [B,C,h,w]; pos_embeds: list of [1,C,h,w].  Returns (memory [B,S,C], shapes, start)."""
        shapes_list = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
        spatial_shapes, level_start_index, ref, normalizer = self._geometry(shapes_list, srcs[0].device)
        src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        from ..ops.colsum import add_channel_vector  # level_embed's gradient from csrc/colsum.hip (ops/colsum.py)
        pos = torch.cat([add_channel_vector(p.flatten(2).transpose(1, 2), self.level_embed[l], -1)
                         for l, p in enumerate(pos_embeds)], 1)
        ref = ref.expand(src.shape[0], -1, -1, -1)
        out, fan = src, None
        n = len(self.encoder.layers)
        for i, layer in enumerate(self.encoder.layers):
            out = layer(out, pos, ref, spatial_shapes, level_start_index, None, normalizer, fan=fan, last=i == n - 1)
            if i < n - 1:
                fan, out = out, None
        return out, spatial_shapes, level_start_index, shapes_list


@SEM_SEG_HEADS_REGISTRY.register()
class MSDeformAttnPixelDecoder(nn.Module):
    """msdeformattn.py:168-359"""

    def __init__(self, input_shape: Dict[str, ShapeSpec], *, transformer_dropout: float, transformer_nheads: int,
                 transformer_dim_feedforward: int, transformer_enc_layers: int, conv_dim: int, mask_dim: int,
                 norm=None, transformer_in_features=("res3", "res4", "res5"), common_stride: int = 4):
        super().__init__()
        transformer_input_shape = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.feature_strides = [v.stride for k, v in input_shape]
        self.feature_channels = [v.channels for k, v in input_shape]
        transformer_input_shape = sorted(transformer_input_shape.items(), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, v in transformer_input_shape]
        transformer_in_channels = [v.channels for k, v in transformer_input_shape]
        self.transformer_feature_strides = [v.stride for k, v in transformer_input_shape]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        chans = transformer_in_channels[::-1] if self.transformer_num_feature_levels > 1 else [transformer_in_channels[-1]]
        self.input_proj = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in chans])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.conv_dim = conv_dim
        self.mask_dim = mask_dim
        self.mask_features = Conv2d(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        c2_xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(np.log2(stride) - np.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[: self.num_fpn_levels]):
            lateral_conv = Conv2d(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=get_norm(norm, conv_dim))
            output_conv = Conv2d(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=get_norm(norm, conv_dim), activation=F.relu)
            c2_xavier_fill(lateral_conv)
            c2_xavier_fill(output_conv)
            self.add_module(f"adapter_{idx + 1}", lateral_conv)
            self.add_module(f"layer_{idx + 1}", output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]

    @classmethod
    def from_config(cls, cfg, input_shape):
        return dict(
            input_shape={k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
            conv_dim=cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
            norm=cfg.MODEL.SEM_SEG_HEAD.NORM, transformer_dropout=cfg.MODEL.MASK_FORMER.DROPOUT,
            transformer_nheads=cfg.MODEL.MASK_FORMER.NHEADS,
            transformer_dim_feedforward=1024,  # hidden constant of the reference (msdeformattn.py:308-309)
            transformer_enc_layers=cfg.MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS,
            transformer_in_features=cfg.MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES,
            common_stride=cfg.MODEL.SEM_SEG_HEAD.COMMON_STRIDE)

    def forward_features(self, features):
        """-> (mask_features [BT,mask_dim,H/4,W/4], out[0], multi_scale_features[3]); fp32 like the reference
        (msdeformattn.py:315: autocast disabled, inputs .float())."""
        from ..ops.linear import forward_precision_scope, range_safe
        with torch.autocast(device_type="cuda", enabled=False), forward_precision_scope(PIXEL_DECODER_FPN_FORWARD):
            srcs, pos = [], []
            for idx, f in enumerate(self.transformer_in_features[::-1]):
                x = features[f].float()
                proj = self.input_proj[idx]  # Sequential(1x1 conv, GroupNorm): the conv is a token-major GEMM (layers.py)
                with range_safe():  # (x: the backbone's own ReLU feature, not a normalised activation)
                    srcs.append(norm_act(conv1x1_or_conv(proj[0], x), proj[1]))
                pos.append(position_embedding_sine(1, x.shape[2], x.shape[3], x.device, self.conv_dim // 2))
            with forward_precision_scope(PIXEL_DECODER_FORWARD):
                y, spatial_shapes, level_start_index, shapes_list = self.transformer(srcs, pos)
            bs = y.shape[0]
            # (one split node: its backward is ONE concatenation - three slices cost three zero-filled [BT, S, C] gradients
            # and two accumulation adds)
            out = [lv.transpose(1, 2).reshape(bs, -1, H, W) for lv, (H, W) in zip(y.split([H * W for H, W in shapes_list], 1), shapes_list)]
            for idx, f in enumerate(self.in_features[: self.num_fpn_levels][::-1]):
                x = features[f].float()
                with range_safe():
                    cur_fpn = self.lateral_convs[idx](x)
                y = upsample_bilinear_add(cur_fpn, out[-1])  # :349-350 cur_fpn + F.interpolate(out[-1]) (one HIP pass for exact 2x)
                out.append(self.output_convs[idx](y))
            multi_scale_features = out[: self.maskformer_num_feature_levels]
            return self.mask_features(out[-1]), out[0], multi_scale_features
