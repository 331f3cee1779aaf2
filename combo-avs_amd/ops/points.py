"""Point sampling (detectron2 `point_sample` == grid_sample(x, 2c-1, bilinear, zeros, align_corners=False)).
HIP kernel: csrc/points.hip (planned: fused with the BCE/dice reductions)."""
import torch
import torch.nn.functional as F


def point_sample(x, coords):
    """x [N,C,H,W], coords [N,P,2] in [0,1] (x,y) -> [N,C,P]"""
    return F.grid_sample(x, (2.0 * coords - 1.0).unsqueeze(2), mode="bilinear", padding_mode="zeros",
                         align_corners=False).squeeze(3)
