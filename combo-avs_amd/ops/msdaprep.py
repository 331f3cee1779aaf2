"""MSDeformAttn prologue (csrc/msdaprep.hip): merged projection row -> sampling locations + softmaxed
attention weights, and the backward of both into one d_proj row."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


class _MsdaPrep(Function):
    @staticmethod
    def forward(ctx, proj, ref, normalizer, M, L, P):
        B, Lq, _ = proj.shape
        loc = torch.empty(B, Lq, M, L, P, 2, device=proj.device, dtype=torch.float32)
        attn = torch.empty(B, Lq, M, L, P, device=proj.device, dtype=torch.float32)
        stride = Lq * L * 2 if ref.shape[0] == B and B > 1 else 0
        _lib.check(_lib.lib().combo_msda_prep_forward_f32(proj.data_ptr(), ref.data_ptr(), normalizer.data_ptr(), B * Lq, Lq, M, L, P,
                                                          stride, loc.data_ptr(), attn.data_ptr(), _lib.current_stream()),
                   "combo_msda_prep_forward_f32")
        ctx.save_for_backward(attn, normalizer)
        ctx.dims = (B, Lq, M, L, P)
        return loc, attn

    @staticmethod
    @once_differentiable
    def backward(ctx, dloc, dattn):
        attn, normalizer = ctx.saved_tensors
        B, Lq, M, L, P = ctx.dims
        dproj = torch.empty(B, Lq, M * L * P * 3, device=attn.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_msda_prep_backward_f32(dloc.contiguous().data_ptr(), dattn.contiguous().data_ptr(), attn.data_ptr(),
                                                           normalizer.data_ptr(), B * Lq, M, L, P, dproj.data_ptr(),
                                                           _lib.current_stream()), "combo_msda_prep_backward_f32")
        return dproj, None, None, None, None, None


def msda_prep(proj, ref, normalizer, n_heads, n_levels, n_points):
    """proj [B,Lq,heads*L*P*3] fp32 (offsets | logits), ref [B or 1,Lq,L,2] (may be batch-expanded), normalizer [L,2]
    (W_l, H_l) -> (sampling_locations [B,Lq,heads,L,P,2], attention_weights [B,Lq,heads,L,P])."""
    if not (proj.is_cuda and ref.is_cuda and normalizer.is_cuda):
        raise RuntimeError("combo_avs_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
    shared = ref.shape[0] == 1 or ref.stride(0) == 0  # one set of reference points for the whole batch
    ref_c = ref[:1].contiguous() if shared else ref.contiguous()
    return _MsdaPrep.apply(proj.contiguous(), ref_c.float(), normalizer.contiguous().float(), n_heads, n_levels, n_points)
