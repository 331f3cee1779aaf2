"""Masked multi-head attention core of the decoder layers on csrc/attention.hip (exact-fp32 MFMA, transposed score tiles).

Replaces `softmax(q k^T / sqrt(d) + mask) v` inside nn.MultiheadAttention as used by CrossAttentionLayer / SelfAttentionLayer
(transformer_decoder/transformer_decoder.py:99-118, 50-58).  Operands are ROW VIEWS [B*L, >= H*32] (last dimension contiguous,
any row stride that is a multiple of 4 floats): the q / k column blocks of a fused projection need no copy.  The mask is one
byte per (frame, query, key) - not replicated over the heads - with rows padded to a multiple of 4 bytes
(ops.masklogit.mask_bits / PackedMask)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def _rows_ok(t):
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0
            and t.data_ptr() % 16 == 0)


# K / V that are column blocks of a merged per-level projection (ops/linear.py memory_kv): key = data_ptr of the k block ->
# (weak reference to the gradient holder - the projection's autograd node owns it -, block index).  The backward pass then writes dk / dv straight into the holder's [rows, blocks * E] buffers
# (csrc/attention.hip strided outputs) instead of into tensors of its own.
kv_gradient_slots = {}


class _Attention(Function):
    @staticmethod
    def forward(ctx, q, k, v, blocked, B, H, bits=None):
        slot = kv_gradient_slots.get(k.data_ptr()) if k.stride(0) != k.shape[1] else None
        holder = slot[0]() if slot is not None else None
        ctx.kv_slot = (holder, slot[1]) if holder is not None else None
        Lq, Lk = q.shape[0] // B, k.shape[0] // B
        E = H * 32
        assert _rows_ok(q) and _rows_ok(k) and _rows_ok(v) and q.shape[1] == E and k.shape[1] == E and v.shape[1] == E
        pitch = 0
        if blocked is not None:
            assert blocked.dtype == torch.uint8 and blocked.is_contiguous() and blocked.shape[:2] == (B, Lq)
            pitch = blocked.shape[2]
            assert pitch % 4 == 0 and pitch >= Lk
        wpitch = 0
        if bits is not None:
            assert bits.dtype == torch.int32 and bits.is_contiguous() and bits.shape[:2] == (B, Lq) and bits.shape[2] * 32 >= Lk
            wpitch = bits.shape[2]
        out = torch.empty(B * Lq, E, device=q.device, dtype=torch.float32)
        lse = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
        scale = 32 ** -0.5
        _lib.check(_lib.lib().combo_attention_forward_f32(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                                          _lib.ptr(blocked), pitch, _lib.ptr(bits), wpitch, B, H, Lq, Lk, scale,
                                                          out.data_ptr(), lse.data_ptr(), _lib.current_stream()),
                   "combo_attention_forward_f32")
        ctx.save_for_backward(q, k, v, blocked, out, lse, bits)
        ctx.dims = (B, H, Lq, Lk, pitch, scale, wpitch)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v, blocked, out, lse, bits = ctx.saved_tensors
        B, H, Lq, Lk, pitch, scale, wpitch = ctx.dims
        E = H * 32
        dout = dout.contiguous()
        dq = torch.empty(B * Lq, E, device=q.device, dtype=torch.float32)
        slot = ctx.kv_slot
        if slot is not None and slot[0].matches(k, v, slot[1]):
            dk, dv = slot[0].blocks(slot[1])  # column blocks of the level's shared gradient buffers
        else:
            dk = torch.empty(B * Lk, E, device=q.device, dtype=torch.float32)
            dv = torch.empty(B * Lk, E, device=q.device, dtype=torch.float32)
        delta = torch.empty(B, H, Lq, device=q.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_attention_backward_ld_f32(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                                              _lib.ptr(blocked), pitch, _lib.ptr(bits), wpitch, B, H, Lq, Lk, scale,
                                                              out.data_ptr(), lse.data_ptr(),
                                                              dout.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dk.stride(0),
                                                              dv.data_ptr(), dv.stride(0), _lib.current_stream()),
                   "combo_attention_backward_ld_f32")
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, blocked, B, H):
    """q [B*Lq, H*32], k / v [B*Lk, H*32] row views, blocked: uint8 [B, Lq, pitch], an ops.masklogit.PackedMask (bytes + the
    bit-packed rows the forward kernel prefers) or None -> [B*Lq, H*32]"""
    if blocked is not None and not torch.is_tensor(blocked):
        return _Attention.apply(q, k, v, blocked.bytes, B, H, blocked.bits)
    return _Attention.apply(q, k, v, blocked, B, H)
