"""Siam-Encoder-Module mix: `channel_weighted_block` (models/utils/misc.py:112-131) and the
per-level mix `f <- f + gate(p) * p` (models/maskformer_model.py:345-352).  HIP kernel: csrc/semmix.hip (planned)."""
import torch
from torch import nn


class channel_weighted_block(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(in_features=dim, out_features=int(dim / 16))
        self.fc2 = nn.Linear(in_features=int(dim / 16), out_features=dim)

    def forward(self, x):
        b, c, _, _ = x.size()
        y = x.float().mean(dim=(2, 3))
        y = torch.sigmoid(self.fc2(torch.relu(self.fc1(y))))
        return y.view(b, c, 1, 1)


def sem_mix(features, pre_sam_features, scale_factor_modules):
    out = {}
    for (key, blk) in zip(features.keys(), scale_factor_modules):
        p = pre_sam_features[key]
        out[key] = features[key] + blk(p).to(p.dtype) * p
    return out
