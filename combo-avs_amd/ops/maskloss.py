"""Fused mask losses of the criterion (csrc/maskloss.hip): importance point selection, BCE + dice, backward."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


@torch.no_grad()
def uncertain_points(masks, mask_index, over_points, extra_points, k):
    """masks [*, h, w] stacked logits (fp32, contiguous), mask_index [NM] int64, over_points [NM,NS,2],
    extra_points [NM,NR,2] or None -> coords [NM, k+NR, 2]"""
    masks = masks.detach()
    _lib.require_cuda(masks, mask_index, over_points, extra_points)
    NM, NS = over_points.shape[:2]
    NR = 0 if extra_points is None else extra_points.shape[1]
    h, w = masks.shape[-2:]
    out = torch.empty(NM, k + NR, 2, device=masks.device, dtype=torch.float32)
    _lib.check(_lib.lib().combo_uncertain_points_f32(masks.data_ptr(), mask_index.data_ptr(), NM, h, w, over_points.data_ptr(), NS,
                                                     _lib.ptr(extra_points), NR, k, out.data_ptr(), _lib.current_stream()),
               "combo_uncertain_points_f32")
    return out


class _MaskLoss(Function):
    @staticmethod
    def forward(ctx, masks, mask_index, gt, gt_index, coords):
        _lib.require_cuda(masks, mask_index, gt, gt_index, coords)
        NM, P = coords.shape[:2]
        h, w = masks.shape[-2:]
        H, W = gt.shape[-2:]
        stats = torch.empty(NM, 4, device=masks.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_mask_loss_forward_f32(masks.data_ptr(), mask_index.data_ptr(), NM, h, w, gt.data_ptr(),
                                                          gt_index.data_ptr(), H, W, coords.data_ptr(), P, stats.data_ptr(),
                                                          _lib.current_stream()), "combo_mask_loss_forward_f32")
        ctx.save_for_backward(masks, mask_index, gt, gt_index, coords, stats)
        bce = stats[:, 0] / P
        dice = 1 - (2 * stats[:, 1] + 1) / (stats[:, 2] + stats[:, 3] + 1)
        return bce, dice

    @staticmethod
    @once_differentiable
    def backward(ctx, g_bce, g_dice):
        masks, mask_index, gt, gt_index, coords, stats = ctx.saved_tensors
        NM, P = coords.shape[:2]
        h, w = masks.shape[-2:]
        H, W = gt.shape[-2:]
        grad = torch.zeros_like(masks)
        _lib.check(_lib.lib().combo_mask_loss_backward_f32(
            masks.data_ptr(), mask_index.data_ptr(), NM, h, w, gt.data_ptr(), gt_index.data_ptr(), H, W, coords.data_ptr(), P,
            stats.data_ptr(), g_bce.contiguous().float().data_ptr(), g_dice.contiguous().float().data_ptr(), grad.data_ptr(), 0,
            None, _lib.current_stream()), "combo_mask_loss_backward_f32")
        return grad, None, None, None, None


def mask_losses(masks, mask_index, gt, gt_index, coords):
    """-> (mean-over-points BCE [NM], dice [NM]) of the matched pairs; differentiable w.r.t. `masks`."""
    return _MaskLoss.apply(masks.contiguous().float(), mask_index, gt.contiguous().float(), gt_index, coords.contiguous())


class _MaskAndCosine(Function):
    """Mask losses of the matched pairs AND the frame-to-frame cosine statistics on ONE stack of mask logits
    `x` [heads, BT, Q, h, w] (the decoder's logits buffer), with ONE gradient buffer: the cosine gradient kernel writes it,
    the mask-loss backward adds its few matched maps in place.  Separately the two terms cost a gathered stack of the GT
    frames, its scatter back (index_add into zero-filled tensors), a stack of the 9 intermediate heads and one 50 MB
    accumulation add per head."""

    @staticmethod
    def forward(ctx, x, cos_heads, n_frame, mask_index, gt, gt_index, coords):
        heads, BT, Q, h, w = x.shape
        lib, st = _lib.lib(), _lib.current_stream()
        NM, P = coords.shape[:2]
        H, W = gt.shape[-2:]
        stats = torch.empty(NM, 4, device=x.device, dtype=torch.float32)
        _lib.check(lib.combo_mask_loss_forward_f32(x.data_ptr(), mask_index.data_ptr(), NM, h, w, gt.data_ptr(), gt_index.data_ptr(),
                                                   H, W, coords.data_ptr(), P, stats.data_ptr(), st), "combo_mask_loss_forward_f32")
        rows, E = cos_heads * BT, Q * h * w
        dot = torch.zeros(rows, device=x.device, dtype=torch.float32)
        nrm = torch.zeros(rows, device=x.device, dtype=torch.float32)
        if rows > 0:
            _lib.check(lib.combo_cosine_stats_f32(x.data_ptr(), rows, E, n_frame, dot.data_ptr(), nrm.data_ptr(), st),
                       "combo_cosine_stats_f32")
        ctx.save_for_backward(x, mask_index, gt, gt_index, coords, stats)
        ctx.meta = (cos_heads, n_frame)
        bce = stats[:, 0] / P
        dice = 1 - (2 * stats[:, 1] + 1) / (stats[:, 2] + stats[:, 3] + 1)
        return bce, dice, dot, nrm

    @staticmethod
    @once_differentiable
    def backward(ctx, g_bce, g_dice, gdot, gnrm):
        x, mask_index, gt, gt_index, coords, stats = ctx.saved_tensors
        cos_heads, n_frame = ctx.meta
        heads, BT, Q, h, w = x.shape
        lib, st = _lib.lib(), _lib.current_stream()
        NM, P = coords.shape[:2]
        H, W = gt.shape[-2:]
        rows, E = cos_heads * BT, Q * h * w
        # the gradient stack is built TRANSPOSED, [BT, heads, Q, h, w], and handed to autograd as the permuted view: that is the
        # layout the mask-logit gradient GEMMs contract over (heads * Q rows per frame, ops/masklogit.py) - without it they
        # start with a 0.5 GB re-layout copy of this very buffer
        grad = torch.empty(BT, heads, Q, h, w, device=x.device, dtype=torch.float32)
        if rows > 0:
            _lib.check(lib.combo_cosine_grad_f32(x.data_ptr(), rows, E, n_frame, gdot.contiguous().float().data_ptr(),
                                                 gnrm.contiguous().float().data_ptr(), grad.data_ptr(), BT, heads, st),
                       "combo_cosine_grad_f32")
        if cos_heads < heads:
            grad[:, cos_heads:].zero_()  # heads without a cosine term (the final prediction)
        grad_index = (mask_index // Q % BT * heads + mask_index // (Q * BT)) * Q + mask_index % Q  # (h, f, q) -> (f, h, q)
        _lib.check(lib.combo_mask_loss_backward_f32(
            x.data_ptr(), mask_index.data_ptr(), NM, h, w, gt.data_ptr(), gt_index.data_ptr(), H, W, coords.data_ptr(), P,
            stats.data_ptr(), g_bce.contiguous().float().data_ptr(), g_dice.contiguous().float().data_ptr(), grad.data_ptr(), 1,
            grad_index.data_ptr(), st), "combo_mask_loss_backward_f32")
        return grad.permute(1, 0, 2, 3, 4), None, None, None, None, None, None


def mask_and_cosine(x, cos_heads, n_frame, mask_index, gt, gt_index, coords):
    """x [heads, BT, Q, h, w] fp32 contiguous -> (bce [NM], dice [NM], dot [cos_heads*BT], nrm [cos_heads*BT])"""
    _lib.require_cuda(x, mask_index, gt, gt_index, coords)
    return _MaskAndCosine.apply(x, cos_heads, n_frame, mask_index, gt.contiguous().float(), gt_index, coords.contiguous())


class _CosineStats(Function):
    """x [rows, E] -> (dot[r] = x_r . x_{r+1} within a clip, nrm[r] = |x_r|^2); csrc/cosine.hip"""

    @staticmethod
    def forward(ctx, x, n_frame):
        _lib.require_cuda(x)
        rows, E = x.shape
        dot = torch.zeros(rows, device=x.device, dtype=torch.float32)
        nrm = torch.zeros(rows, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_cosine_stats_f32(x.data_ptr(), rows, E, n_frame, dot.data_ptr(), nrm.data_ptr(),
                                                     _lib.current_stream()), "combo_cosine_stats_f32")
        ctx.save_for_backward(x)
        ctx.n_frame = n_frame
        return dot, nrm

    @staticmethod
    @once_differentiable
    def backward(ctx, gdot, gnrm):
        x, = ctx.saved_tensors
        rows, E = x.shape
        grad = torch.empty_like(x)
        _lib.check(_lib.lib().combo_cosine_grad_f32(x.data_ptr(), rows, E, ctx.n_frame, gdot.contiguous().float().data_ptr(),
                                                    gnrm.contiguous().float().data_ptr(), grad.data_ptr(), 0, 0, _lib.current_stream()),
                   "combo_cosine_grad_f32")
        return grad, None


def cosine_stats(x, n_frame):
    return _CosineStats.apply(x.contiguous().float(), n_frame)
