"""`mask_embed @ pixel_embed` + next-layer attention mask (transformer_decoder.py:498-507 of the reference).
HIP kernels: csrc/masklogit.hip (MFMA contraction with fused bilinear-downsample + sigmoid<0.5 epilogue)."""
import torch
import torch.nn.functional as F


def mask_logits_and_attn_mask(mask_embed, mf_tok, hw, target_size):
    """mask_embed [BT,Q,C], mf_tok [BT,HW,C] token-major -> (logits [BT,Q,H,W], blocked bool [BT,Q,h*w])."""
    bt, Q, _ = mask_embed.shape
    logits = torch.bmm(mask_embed, mf_tok.transpose(1, 2)).view(bt, Q, hw[0], hw[1])
    with torch.no_grad():
        am = F.interpolate(logits, size=target_size, mode="bilinear", align_corners=False)
        blocked = (am.sigmoid() < 0.5).flatten(2)
    return logits, blocked
