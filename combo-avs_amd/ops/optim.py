"""Fused clip + AdamW on a flat parameter segment (HIP kernel csrc/optim.hip on the GPU).  The CPU branch exists
only so that the world_size-2 gloo tests can exercise the data-parallel host logic in a GPU-less container; it is
never taken for CUDA tensors."""
import torch

from .. import _lib


def adamw_segment(p, g, m, v, clip_coef, lr, wd, b1, b2, eps, bc1, bc2):
    """torch.optim.AdamW semantics: p *= 1 - lr*wd; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)."""
    if p.is_cuda:
        _lib.require_cuda(p, g, m, v, clip_coef)
        if g.dtype != torch.float32:
            g = g.float()
        _lib.check(_lib.lib().combo_adamw_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                              clip_coef.data_ptr(), lr, wd, b1, b2, eps, bc1, bc2, _lib.current_stream()),
                   "combo_adamw_f32")
        return
    g = g.float() * clip_coef
    if wd != 0:
        p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
