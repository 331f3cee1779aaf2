"""`mask_embed @ pixel_embed` + next-layer attention mask (transformer_decoder.py:498-507 of the reference).
HIP kernels: the forward contraction of every prediction head is a batched exact-fp32 MFMA GEMM (csrc/gemm_f32.hip, one
problem per frame: [Q,C] x [HW,C]^T - its output is thresholded at 0 into the next layer's attention mask, so it computes in
true fp32); the mask itself comes from csrc/maskbits.hip (the same contraction against the DOWNSAMPLED pixel embedding, its MFMA
result balloted into the bit-packed rows); the two gradient GEMMs run ONCE over the concatenated heads on the 3-product bf16 kernels (csrc/gemm_nt3.hip
batched, csrc/gemm_tn.hip grouped, one problem per frame)."""
import ctypes

import torch

from .. import _lib


def _hip_ok(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts)


def presplit_batched(x, transpose, f16=False):
    """hi/lo image (csrc/gemm_nt3.hip; bf16 pieces, or fp16 ones for the "f16x3" forward mode) of a contiguous [B, R, C] tensor:
    [B, R, C] (K = C) or, transposed, [B, C, R]."""
    from .linear import split_pieces
    B, R, C = x.shape
    N, K = (C, R) if transpose else (R, C)
    img = torch.empty(B, N, K, device=x.device, dtype=torch.float32)
    ld_row, ld_col = (1, C) if transpose else (C, 1)
    with split_pieces(f16):
        _lib.check(_lib.lib().combo_presplit_bf16x2_batched_f32(x.data_ptr(), ld_row, ld_col, R * C, N, K, B, img.data_ptr(),
                                                                _lib.current_stream()), "combo_presplit_bf16x2_batched_f32")
    return img


def gemm_nt_batched(a, img, out):
    """out[b] = a[b] @ B[b]^T for contiguous a [B, M, K], pre-split image [B, N, K], out [B, M, N]"""
    B, M, K = a.shape
    N = img.shape[1]
    with _lib.timed("gemm_nt_x3", (B * M, N, K)):
        rc = _lib.lib().combo_gemm_nt_x3_pre_batched_f32(a.data_ptr(), K, M * K, img.data_ptr(), N * K, out.data_ptr(), N, M * N,
                                                         M, N, K, B, 0, _lib.current_stream())
    _lib.check(rc, "combo_gemm_nt_x3_pre_batched_f32")
    return out


class PackedMask:
    """The attention mask of one decoder layer in the two forms csrc/attention.hip reads: `bytes` uint8 [BT,Q,pitch]
    (1 = masked out, the backward kernels) and `bits` int32 [BT,Q,wpitch] (bit k of word j = key 32 j + k, the forward
    kernel: one word per query and 32-key tile)."""

    def __init__(self, bytes_, bits):
        self.bytes, self.bits = bytes_, bits


def downsample_tokens(mf_tok, hw_in, hw_out):
    """mf_tok [BT, H*W, C] (token-major mask features) -> [BT, h*w, C]: F.interpolate(..., mode="bilinear",
    align_corners=False) of transformer_decoder.py:502 applied to the pixel embedding (it commutes with the contraction)."""
    x = mf_tok.detach()
    _lib.require_cuda(x)
    BT, HW, C = x.shape
    (H, W), (h, w) = hw_in, hw_out
    assert H * W == HW and x.is_contiguous() and x.dtype == torch.float32
    out = torch.empty(BT, h * w, C, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().combo_downsample_tokens_f32(x.data_ptr(), BT, H, W, h, w, C, out.data_ptr(), _lib.current_stream()),
               "combo_downsample_tokens_f32")
    return out


def mask_bits(mask_embed, mfd, reset_full_rows=True, with_bytes=False):
    """The attention mask of a decoder layer, fused (csrc/maskbits.hip): mask_embed [BT, Q, 256] x downsampled pixel embedding
    mfd [BT, hw, 256] -> PackedMask (bit-packed rows; byte rows only on request).  No score tensor is written."""
    me, mfd = mask_embed.detach(), mfd.detach()
    _lib.require_cuda(me, mfd)
    BT, Q, C = me.shape
    hw = mfd.shape[1]
    if not (_hip_ok(me, mfd) and C == 256 and mfd.shape[2] == C and hw <= 4096):
        raise RuntimeError("mask_bits: fp32 contiguous CUDA tensors [BT, Q, 256] / [BT, hw <= 4096, 256] expected")
    wpitch = (hw + 63) // 64 * 2
    pitch = (hw + 3) // 4 * 4
    bits = torch.empty(BT, Q, wpitch, dtype=torch.int32, device=me.device)
    by = torch.empty(BT, Q, pitch, dtype=torch.uint8, device=me.device) if with_bytes else None
    _lib.check(_lib.lib().combo_mask_bits_f32(me.data_ptr(), mfd.data_ptr(), BT, Q, hw, C, 1 if reset_full_rows else 0, wpitch,
                                              bits.data_ptr(), pitch, _lib.ptr(by), _lib.current_stream()), "combo_mask_bits_f32")
    return PackedMask(by, bits)


def mask_logits_all_into(mask_embeds, mf_tok, out):
    """no-grad: out[h] = mask_embeds[h] @ mf_tok^T for ALL prediction heads in one launch (exact fp32, csrc/gemm_f32.hip);
    mask_embeds: list of [BT, Q, C]; out [heads, BT, Q, HW]"""
    from . import linear as L
    mf = mf_tok.detach()
    if L.FORWARD_PRECISION != "fp32":  # the head's bf16 throughput mode: per head, one bf16 product per multiply-add
        img = presplit_batched(mf, transpose=False, f16=L.forward_f16())
        lib = _lib.lib()
        prev = lib.combo_gemm_nt2_products(L.forward_products())
        try:
            for h, m in enumerate(mask_embeds):
                gemm_nt_batched(m.detach().contiguous(), img, out[h])
        finally:
            lib.combo_gemm_nt2_products(prev)
        return out
    me = torch.stack([m.detach() for m in mask_embeds])  # [heads, BT, Q, C]
    heads, BT, Q, C = me.shape
    HW = mf.shape[1]
    if not (_hip_ok(me, mf, out) and C % 16 == 0 and Q * HW * 4 < 2 ** 31 - 1):
        raise RuntimeError("mask_logits_all_into: fp32 contiguous 16-byte aligned CUDA tensors with C % 16 == 0 expected")
    with _lib.timed("gemm_nt_f32", (heads * BT * Q, HW, C)):
        rc = _lib.lib().combo_mask_logits_all_f32(me.data_ptr(), mf.data_ptr(), out.data_ptr(), heads, BT, Q, HW, C,
                                                  _lib.current_stream())
    _lib.check(rc, "combo_mask_logits_all_f32")
    return out


def pack_mask(blocked, reset_full_rows=True):
    """bool [BT,Q,n] (True = masked out; as produced by a prediction head) -> PackedMask in the layout mask_bits
    writes, row reset of :458 applied.  Host-side torch ops: for tests that inject masks, not on the step's path."""
    bt, Q, n = blocked.shape
    b = blocked.clone()
    if reset_full_rows:
        b[b.all(-1)] = False
    pitch, wpitch = (n + 3) // 4 * 4, (n + 63) // 64 * 2
    by = torch.ones(bt, Q, pitch, dtype=torch.uint8, device=b.device)
    by[:, :, :n] = b.to(torch.uint8)
    full = torch.ones(bt, Q, wpitch * 32, dtype=torch.int64, device=b.device)
    full[:, :, :n] = b.to(torch.int64)
    words = (full.view(bt, Q, wpitch, 32) << torch.arange(32, device=b.device)).sum(-1)  # bit k of word j = key 32 j + k
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    return PackedMask(by.contiguous(), words.contiguous())


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
        # [BT, heads * Q, HW]: a free view when the criterion hands the stack over transposed (ops/maskloss.py builds its
        # gradient as [BT, heads, ...]); a 0.5 GB re-layout copy for a gradient in the forward layout
        dL2 = dL.permute(1, 0, 2, 3).reshape(bt, nh * Q, HW)
        ME = torch.cat(mes, dim=1)  # [BT, nh*Q, C]
        C = mf_tok.shape[2]
        if _hip_ok(dL2, mf_tok, ME) and HW % 16 == 0 and C % 4 == 0 and HW % 4 == 0 and nh * Q >= 256 and C >= 64:
            # dME[b] = dL[b] . MF[b]: batched NT GEMM against the per-frame transposed image of the mask features
            dME = gemm_nt_batched(dL2, presplit_batched(mf_tok, transpose=True), torch.empty(bt, nh * Q, C, device=dL.device))
            # dMF[b] = dL[b]^T . ME[b]: reduction over the heads*Q rows = the weight-gradient kernel, one problem per frame
            from .linear import _TnProblem
            dMF = torch.empty(bt, HW, C, device=dL.device, dtype=torch.float32)
            prob = (_TnProblem * bt)()
            for b in range(bt):
                prob[b] = _TnProblem(dL2[b].data_ptr(), ME[b].data_ptr(), dMF[b].data_ptr(), 0, HW, C, nh * Q, HW, C, 1)
            with _lib.timed("gemm_tn_x3_grouped", (2.0 * bt * nh * Q * HW * C, bt)):
                rc = _lib.lib().combo_gemm_tn_x3_grouped_f32(ctypes.cast(prob, ctypes.c_void_p), bt, _lib.current_stream())
            _lib.check(rc, "combo_gemm_tn_x3_grouped_f32")
            return (dMF, None) + tuple(dME[:, i * Q:(i + 1) * Q] for i in range(nh))
        _lib.fallback_notice("ops.masklogit._MaskLogitsAll.backward", f"HW {HW}, C {C}, heads*Q {nh * Q}: the batched 3-product "
                             "kernels take HW % 16 == 0, C % 4 == 0, C >= 64, heads*Q >= 256")
        with torch.autocast("cuda", enabled=False):
            dME = torch.bmm(dL2, mf_tok)  # [BT, nh*Q, C]
            dMF = torch.bmm(dL2.transpose(1, 2), ME)  # [BT, HW, C]
        return (dMF, None) + tuple(dME[:, i * Q:(i + 1) * Q] for i in range(nh))


def attach_mask_logit_grads(mf_tok, buffer, mask_embeds):
    """buffer [heads, BT, Q, HW] (values already computed) -> same values, differentiable w.r.t. mf_tok / mask_embeds"""
    return _MaskLogitsAll.apply(mf_tok, buffer, *mask_embeds)


def mask_logits_into(mask_embed, mf_tok, out):
    """no-grad: out[BT,Q,HW] = mask_embed @ mf_tok^T in exact fp32 (csrc/gemm_f32.hip, batched over the frames)"""
    me, mf = mask_embed.detach(), mf_tok.detach()
    B, Q, C = me.shape
    HW = mf.shape[1]
    if not (_hip_ok(me, mf, out) and C % 16 == 0 and Q * HW * 4 < 2 ** 31 - 1):
        raise RuntimeError("mask_logits_into: fp32 contiguous 16-byte aligned CUDA tensors with C % 16 == 0 expected")
    with _lib.timed("gemm_nt_f32", (B * Q, HW, C)):
        rc = _lib.lib().combo_gemm_nt_batched_f32(me.data_ptr(), C, Q * C, mf.data_ptr(), C, HW * C, out.data_ptr(), HW, Q * HW,
                                                  Q, HW, C, B, 0, _lib.current_stream())
    _lib.check(rc, "combo_gemm_nt_batched_f32")
    return out
