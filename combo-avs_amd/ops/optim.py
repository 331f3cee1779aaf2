"""Fused clip + AdamW on a flat parameter segment.  HIP kernel: csrc/optim.hip (planned); elementwise torch ops on
the flat buffers until then (device-agnostic so the gloo CPU tests exercise the same host logic)."""
import torch


def adamw_segment(p, g, m, v, clip_coef, lr, wd, b1, b2, eps, bc1, bc2):
    """torch.optim.AdamW semantics: p *= 1 - lr*wd; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)."""
    g = g.float() * clip_coef
    if wd != 0:
        p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
