"""Siam-Encoder-Module mix: `channel_weighted_block` (models/utils/misc.py:112-131) and the
per-level mix `f <- f + gate(p) * p` (models/maskformer_model.py:345-352).  The global average pool and the mix run on
csrc/semmix.hip (channels-last, bf16 or fp32 in, fp32 out); the two tiny gate GEMVs run on the head's dense-layer kernels (ops/linear.py)."""
import torch
from torch import nn

from ..ops import linear as L
from ..ops import semmix as K


class channel_weighted_block(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(in_features=dim, out_features=int(dim / 16))
        self.fc2 = nn.Linear(in_features=int(dim / 16), out_features=dim)

    def gate(self, y):
        """[B, C] pooled features -> [B, C] channel weights"""
        return torch.sigmoid(L.linear(L.linear(y, self.fc1.weight, self.fc1.bias, relu=True), self.fc2.weight, self.fc2.bias))

    def forward(self, x):
        b, c, _, _ = x.size()
        y = K.global_avg_pool(x)  # [B,C] fp32
        # the two gate GEMVs ([BT, C] rows): the head's exact-fp32 kernels (a BLAS tile GEMM takes 65 us for these shapes)
        y = torch.sigmoid(L.linear(L.linear(y, self.fc1.weight, self.fc1.bias, relu=True), self.fc2.weight, self.fc2.bias))
        return y.view(b, c, 1, 1)


def sem_mix(features, pre_sam_features, scale_factor_modules):
    out = {}
    for (key, blk) in zip(features.keys(), scale_factor_modules):
        p = pre_sam_features[key]
        if p.is_cuda and isinstance(blk, channel_weighted_block):
            # pool + gate + mix as one autograd node: the pool's gradient rides in the mix-backward kernel (ops/semmix.py _GateMix)
            out[key] = K.gate_mix(features[key], p, blk.gate, tuple(blk.parameters()))
        else:
            out[key] = K.mix(features[key], p, blk(p).view(p.shape[0], p.shape[1]))
    return out
