"""Spatial-reduction attention of the PVTv2 backbone on csrc/sra_attention.hip (bf16 MFMA, fp32 accumulation).

Replaces `F.scaled_dot_product_attention` in backbone_pvt.Attention (reference: models/modeling/backbone/pvtv2.py:104-118:
`attn = (q @ k.transpose(-2, -1)) * self.scale; attn = attn.softmax(dim=-1); x = (attn @ v).transpose(1, 2).reshape(B, N, C)`)
for head dimension 64 and <= 256 keys - every stage of PVTv2-B5 at 224 x 224 (49 keys) and 512 x 512 (256 keys).  The kernels read
the q / kv projections' outputs where they lie and write the [B, N, C] tensor the output projection consumes: no head
transposes, no `unbind` stack in the backward pass (dkv comes back as ONE tensor in the kv projection's layout)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib

ENABLED = True  # module constant (tests / tools flip it): False = the library's scaled_dot_product_attention


def usable(q, kv, num_heads):
    """q [B, N, C], kv [B, Nk, 2 C] bf16 CUDA tensors of a geometry the kernels take"""
    if not (ENABLED and q.is_cuda and q.dtype == torch.bfloat16 and kv.dtype == torch.bfloat16 and q.dim() == 3 and kv.dim() == 3):
        return False
    C = q.shape[2]
    if C % num_heads or kv.shape[2] != 2 * C or kv.shape[0] != q.shape[0]:
        return False
    return bool(_lib.lib().combo_sra_attention_ok(q.shape[1], kv.shape[1], C // num_heads))


class _SraAttention(Function):
    @staticmethod
    def forward(ctx, q, kv, num_heads, scale):
        q, kv = q.contiguous(), kv.contiguous()
        _lib.require_cuda(q, kv)
        B, N, C = q.shape
        Nk = kv.shape[1]
        out = torch.empty_like(q)
        lse2 = torch.empty(B, num_heads, (N + 31) // 32 * 32, device=q.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_sra_attention_forward_bf16(q.data_ptr(), kv.data_ptr(), out.data_ptr(), lse2.data_ptr(), B, N, Nk,
                                                              num_heads, float(scale), _lib.current_stream()),
                   "combo_sra_attention_forward_bf16")
        ctx.save_for_backward(q, kv, out, lse2)
        ctx.num_heads, ctx.scale = num_heads, float(scale)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, kv, out, lse2 = ctx.saved_tensors
        dout = dout.contiguous()
        B, N, C = q.shape
        Nk = kv.shape[1]
        lib = _lib.lib()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        delta = torch.empty_like(lse2)
        part = torch.empty(int(lib.combo_sra_attention_backward_workspace(B, N, Nk, ctx.num_heads)), device=q.device, dtype=torch.float32)
        _lib.check(lib.combo_sra_attention_backward_bf16(q.data_ptr(), kv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse2.data_ptr(),
                                                         delta.data_ptr(), part.data_ptr(), dq.data_ptr(), dkv.data_ptr(), B, N, Nk,
                                                         ctx.num_heads, ctx.scale, _lib.current_stream()),
                   "combo_sra_attention_backward_bf16")
        return dq, dkv, None, None


def sra_attention(q, kv, num_heads, scale):
    """q [B, N, C] (the q projection's output), kv [B, Nk, 2 C] (the kv projection's output: per key [k of all heads | v of all
    heads]) -> softmax(q k^T scale) v as [B, N, C]"""
    return _SraAttention.apply(q, kv, num_heads, scale)
