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


def mask_logits_and_attn_mask(mask_embed, mf_tok, hw, target_size):
    """mask_embed [BT,Q,C], mf_tok [BT,HW,C] token-major -> (logits [BT,Q,H,W], blocked bool [BT,Q,h*w]).
    The returned mask already has fully-blocked rows reset (it is only ever consumed by the next layer)."""
    bt, Q, _ = mask_embed.shape
    with torch.autocast("cuda", enabled=False):  # mask logits are always produced in fp32
        logits = torch.bmm(mask_embed.float(), mf_tok.float().transpose(1, 2)).view(bt, Q, hw[0], hw[1])
    blocked = attn_mask(logits.detach().float().contiguous(), target_size, True)
    return logits, blocked
