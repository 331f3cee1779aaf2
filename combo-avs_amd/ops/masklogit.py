"""`mask_embed @ pixel_embed` + next-layer attention mask (transformer_decoder.py:498-507 of the reference).
HIP kernels: csrc/attnmask.hip (bilinear-downsample + sigmoid<0.5 + full-row reset in one launch) and
csrc/masklogit.hip (fp32-MFMA contraction)."""
import torch

from .. import _lib


def attn_mask(logits, target_size, reset_full_rows=True):
    """logits [BT,Q,H,W] fp32 -> blocked bool [BT,Q,h*w] (True = masked out), row reset of :458 applied."""
    _lib.require_cuda(logits)
    bt, Q, H, W = logits.shape
    h, w = target_size
    out = torch.empty((bt, Q, h * w), dtype=torch.uint8, device=logits.device)
    _lib.check(_lib.lib().combo_attn_mask_f32(logits.data_ptr(), bt * Q, H, W, h, w, 1 if reset_full_rows else 0,
                                              out.data_ptr(), _lib.current_stream()), "combo_attn_mask_f32")
    return out.view(torch.bool)


class _MaskLogitsAll(torch.autograd.Function):
    """All prediction heads' mask logits as ONE autograd node.

    The decoder needs head i's logits before layer i runs (they gate its cross-attention), so the forward values are
    produced head by head (no grad) into slices of one [heads, BT, Q, HW] buffer.  This node only attaches the
    gradient: with L = ME . MF^T per frame, dME = dL . MF and dMF = dL^T . ME are evaluated ONCE over the
    concatenated heads (K = heads*Q) instead of 2 GEMMs + a 128 MB accumulation per head."""

    @staticmethod
    def forward(ctx, mf_tok, buffer, *mask_embeds):
        ctx.save_for_backward(mf_tok, *mask_embeds)
        return buffer

    @staticmethod
    def backward(ctx, dL):
        mf_tok, *mes = ctx.saved_tensors
        nh, bt, Q, HW = dL.shape
        dL2 = dL.permute(1, 0, 2, 3).reshape(bt, nh * Q, HW)  # one 50 MB re-layout instead of 9 x 128 MB accumulations
        ME = torch.cat(mes, dim=1)  # [BT, nh*Q, C]
        with torch.autocast("cuda", enabled=False):
            dME = torch.bmm(dL2, mf_tok)  # [BT, nh*Q, C]
            dMF = torch.bmm(dL2.transpose(1, 2), ME)  # [BT, HW, C]
        return (dMF, None) + tuple(dME[:, i * Q:(i + 1) * Q] for i in range(nh))


def attach_mask_logit_grads(mf_tok, buffer, mask_embeds):
    """buffer [heads, BT, Q, HW] (values already computed) -> same values, differentiable w.r.t. mf_tok / mask_embeds"""
    return _MaskLogitsAll.apply(mf_tok, buffer, *mask_embeds)


def mask_logits_into(mask_embed, mf_tok, out):
    """no-grad: out[BT,Q,HW] = mask_embed @ mf_tok^T (fp32)"""
    with torch.no_grad(), torch.autocast("cuda", enabled=False):
        torch.bmm(mask_embed.detach().float(), mf_tok.detach().float().transpose(1, 2), out=out)
    return out


def mask_logits_and_attn_mask(mask_embed, mf_tok, hw, target_size):
    """mask_embed [BT,Q,C], mf_tok [BT,HW,C] token-major -> (logits [BT,Q,H,W], blocked bool [BT,Q,h*w]).
    The returned mask already has fully-blocked rows reset (it is only ever consumed by the next layer)."""
    bt, Q, _ = mask_embed.shape
    with torch.autocast("cuda", enabled=False):  # mask logits are always produced in fp32
        logits = torch.bmm(mask_embed.float(), mf_tok.float().transpose(1, 2)).view(bt, Q, hw[0], hw[1])
    blocked = attn_mask(logits.detach().float().contiguous(), target_size, True)
    return logits, blocked
