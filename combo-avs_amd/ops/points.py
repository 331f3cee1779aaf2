"""Point sampling (detectron2 `point_sample` == grid_sample(x, 2c-1, bilinear, zeros, align_corners=False)) for the
reference-shaped single-call methods kept for API parity (`SetCriterion.loss_masks`, `HungarianMatcher.forward` on one decoder
output: criterion.py:137-186, matcher.py:84-136).  The training step does not come through here: `SetCriterion._losses` samples
inside the fused kernels (csrc/matcher.hip: cost matrices; csrc/maskloss.hip: importance sampling + BCE / dice)."""
import torch
import torch.nn.functional as F


def point_sample(x, coords):
    """x [N,C,H,W], coords [N,P,2] in [0,1] (x,y) -> [N,C,P]"""
    return F.grid_sample(x, (2.0 * coords - 1.0).unsqueeze(2), mode="bilinear", padding_mode="zeros",
                         align_corners=False).squeeze(3)
