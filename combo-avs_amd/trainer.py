"""Data-parallel training step for the COMBO hot path on one MI355X node.

One process per GPU.  All trainable parameters live in ONE flat fp32 buffer and all gradients in ONE flat fp32
buffer (parameters / .grad are views), so a step issues exactly one RCCL all-reduce over xGMI (plus the 4-byte
`num_masks` all-reduce inside the criterion, criterion.py:263-265 of the reference), one global-norm clip and one
AdamW pass.  Semantics follow train_net.py:147-226 of the reference:
  * AdamW(lr = BASE_LR, weight_decay = WEIGHT_DECAY), lr x BACKBONE_MULTIPLIER for parameters of modules whose
    name contains "backbone" (this matches `pre_sam_backbone` too, train_net.py:182),
  * weight_decay = WEIGHT_DECAY_NORM (0) for norm-layer parameters, WEIGHT_DECAY_EMBED (0) for nn.Embedding,
  * full-model gradient-norm clipping to CLIP_VALUE (0.01) before the update (train_net.py:205-211),
  * gradients are averaged over ranks (DDP semantics).
Parameters are laid out grouped by (lr, weight_decay) so every group is one contiguous segment.
"""
import math

import torch
import torch.distributed as dist
from torch import nn

NORM_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.GroupNorm, nn.InstanceNorm1d,
              nn.InstanceNorm2d, nn.InstanceNorm3d, nn.LayerNorm, nn.LocalResponseNorm)


def param_groups(model, base_lr, weight_decay, backbone_multiplier=0.1, weight_decay_norm=0.0, weight_decay_embed=0.0):
    """-> list of (param, name, lr, wd) following train_net.py:170-194 of the reference."""
    out, memo = [], set()
    for module_name, module in model.named_modules():
        for pname, p in module.named_parameters(recurse=False):
            if not p.requires_grad or p in memo:
                continue
            memo.add(p)
            lr, wd = base_lr, weight_decay
            if "backbone" in module_name:
                lr = lr * backbone_multiplier
            if "relative_position_bias_table" in pname or "absolute_pos_embed" in pname:
                wd = 0.0
            if isinstance(module, NORM_TYPES):
                wd = weight_decay_norm
            if isinstance(module, nn.Embedding):
                wd = weight_decay_embed
            out.append((p, f"{module_name}.{pname}" if module_name else pname, lr, wd))
    return out


class FlatAdamW:
    """Flat-buffer AdamW with full-model grad-norm clipping and a single gradient all-reduce."""

    def __init__(self, model, base_lr=1e-4, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay_norm=0.0, weight_decay_embed=0.0, grad_dtype=torch.float32):
        entries = param_groups(model, base_lr, weight_decay, backbone_multiplier, weight_decay_norm, weight_decay_embed)
        entries.sort(key=lambda e: (e[2], e[3]))  # stable: contiguous (lr, wd) segments
        self.entries = entries
        # every (lr, wd) segment starts on a 16-byte boundary (vectorised fused AdamW kernel)
        # ... and so does every parameter (the dense-layer kernels read weights with 16-byte loads)
        total = sum((p.numel() + 3) // 4 * 4 for p, _, _, _ in entries)
        dev = entries[0][0].device
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=grad_dtype, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.segments = []  # (start, end, lr, wd)
        self.grad_views = []
        self.params = [e[0] for e in entries]
        off = 0
        self.flat_param.zero_()
        for p, name, lr, wd in entries:
            n = p.numel()
            off = (off + 3) // 4 * 4
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p)
            self.grad_views.append(self.flat_grad[off:off + n].view_as(p))
            if self.segments and self.segments[-1][2] == lr and self.segments[-1][3] == wd:
                self.segments[-1][1] = off + n
            else:
                self.segments.append([off, off + n, lr, wd])
            off += n
        self.betas, self.eps, self.clip_value = betas, eps, clip_value
        self.step_count = 0
        self.lr_scale = 1.0  # WarmupPolyLR factor, set by the caller each iteration
        self.numel = total

    def zero_grad(self):
        self.flat_grad.zero_()

    def backward(self, loss):
        """d loss / d params straight into the flat gradient buffer: `autograd.grad` (no per-parameter AccumulateGrad
        add kernels - 530 launches/step for this model) followed by one multi-tensor copy."""
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        dst = [v for v, g in zip(self.grad_views, grads) if g is not None]
        src = [g for g in grads if g is not None]
        if len(src) != len(grads):
            self.flat_grad.zero_()
        torch._foreach_copy_(dst, src)

    def all_reduce_grads(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_grad)  # ONE collective (RCCL over xGMI on GPU; gloo in the CPU tests)
            self.flat_grad.div_(dist.get_world_size())

    @torch.no_grad()
    def step(self):
        from .ops import optim
        self.step_count += 1
        g = self.flat_grad
        if self.clip_value and self.clip_value > 0:
            total_norm = torch.linalg.vector_norm(g.float())  # clip_grad_norm_(all params, 0.01)
            clip_coef = torch.clamp(self.clip_value / (total_norm + 1e-6), max=1.0)
        else:
            clip_coef = torch.ones((), device=g.device)
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        for s, e, lr, wd in self.segments:
            optim.adamw_segment(self.flat_param[s:e], g[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], clip_coef,
                                lr * self.lr_scale, wd, b1, b2, self.eps, bc1, bc2)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}


def poly_lr_factor(it, max_iter, power=0.9, constant_ending=0.0):
    """detectron2 projects/deeplab WarmupPolyLR with WARMUP_ITERS = 0 (train_net.py:140-145)."""
    f = math.pow(1.0 - it / max_iter, power)
    return f if f >= constant_ending or constant_ending <= 0 else constant_ending


def train_step(model, optimizer, batched_inputs):
    """forward -> 39-term loss -> backward -> one all-reduce -> clip + AdamW.  Returns the loss dict (device tensors)."""
    loss_dict = model(batched_inputs)
    total = torch.stack(list(loss_dict.values())).sum()
    optimizer.backward(total)
    optimizer.all_reduce_grads()
    optimizer.step()
    return loss_dict
