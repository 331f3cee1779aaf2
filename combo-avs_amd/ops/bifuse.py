"""Token stage of the bilateral audio-visual fusion (LayerNorm -> scores -> softmax over HW -> rank-8 update +
attention pooling), see modeling/fusion.py for the algebra.  HIP kernels: csrc/bifuse.hip (3 launches forward,
2 backward), called through the C ABI; no PyTorch fallback."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib

_seed_counter = [0]
_step_counters = {}


def step_counter(device):
    """Device-resident 64-bit step counter that the kernels mix into the Philox key (`seed_step` of the C ABI).  A
    captured training step (trainer.GraphedTrainStep) increments it inside the hipGraph, so every replay draws fresh
    dropout masks although the kernel arguments are frozen."""
    key = str(device)
    if key not in _step_counters:
        _step_counters[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return _step_counters[key]


class _BifuseTokenOp(Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, pos, u, c, z, b_ov, gamma_v, p_drop, drop_v, drop_a, seed):
        _lib.require_cuda(x, ln_w, ln_b, pos, u, c, z, b_ov, gamma_v, drop_v, drop_a)
        lib = _lib.lib()
        B, N, C = x.shape
        H = u.shape[1]
        chunks = lib.combo_bifuse_chunks(B, N)
        dev, f32 = x.device, torch.float32
        y = torch.empty_like(x)
        scores = torch.empty(B, H, N, device=dev, dtype=f32)
        stat = torch.empty(B, H, 2, device=dev, dtype=f32)
        part_ws = torch.empty(B, chunks, H, 2, device=dev, dtype=f32)
        pooled_part = torch.empty(B, chunks, H, C, device=dev, dtype=f32)
        spa_part = torch.empty(B, chunks, H, device=dev, dtype=f32)
        _lib.check(lib.combo_bifuse_forward_f32(
            x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), eps, pos.data_ptr(), u.data_ptr(), c.data_ptr(), z.data_ptr(),
            b_ov.data_ptr(), gamma_v.data_ptr(), _lib.ptr(drop_v), _lib.ptr(drop_a), p_drop, seed, step_counter(x.device).data_ptr(), B, N, C, H,
            y.data_ptr(), scores.data_ptr(), stat.data_ptr(), part_ws.data_ptr(), pooled_part.data_ptr(),
            spa_part.data_ptr(), _lib.current_stream()), "combo_bifuse_forward_f32")
        ctx.save_for_backward(x, ln_w, ln_b, pos, u, z, b_ov, gamma_v, scores, stat, drop_v, drop_a)
        ctx.meta = (eps, p_drop, seed, chunks)
        return y, pooled_part.sum(1), spa_part.sum(1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dpooled, dspa):
        x, ln_w, ln_b, pos, u, z, b_ov, gamma_v, scores, stat, drop_v, drop_a = ctx.saved_tensors
        eps, p_drop, seed, chunks = ctx.meta
        lib = _lib.lib()
        B, N, C = x.shape
        H = u.shape[1]
        dev, f32 = x.device, torch.float32
        dy, dpooled, dspa = dy.contiguous().float(), dpooled.contiguous().float(), dspa.contiguous().float()
        dp = torch.empty(B, H, N, device=dev, dtype=f32)
        r_part = torch.empty(B, chunks, H, device=dev, dtype=f32)
        dz_part = torch.empty(B, chunks, H, C, device=dev, dtype=f32)
        dgb_part = torch.empty(B, chunks, 2, C, device=dev, dtype=f32)
        st = _lib.current_stream()
        _lib.check(lib.combo_bifuse_backward1_f32(
            x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), eps, scores.data_ptr(), stat.data_ptr(), z.data_ptr(),
            b_ov.data_ptr(), gamma_v.data_ptr(), _lib.ptr(drop_v), _lib.ptr(drop_a), p_drop, seed, step_counter(x.device).data_ptr(), dy.data_ptr(),
            dpooled.data_ptr(), dspa.data_ptr(), B, N, C, H, dp.data_ptr(), r_part.data_ptr(), dz_part.data_ptr(),
            dgb_part.data_ptr(), st), "combo_bifuse_backward1_f32")
        rtot = r_part.sum(1)
        dx = torch.empty_like(x)
        du_part = torch.empty(B, chunks, H, C, device=dev, dtype=f32)
        dc_part = torch.empty(B, chunks, H, device=dev, dtype=f32)
        dln_part = torch.empty(B, chunks, 2, C, device=dev, dtype=f32)
        _lib.check(lib.combo_bifuse_backward2_f32(
            x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), eps, pos.data_ptr(), scores.data_ptr(), stat.data_ptr(),
            u.data_ptr(), _lib.ptr(drop_a), p_drop, seed, step_counter(x.device).data_ptr(), dy.data_ptr(), dpooled.data_ptr(),
            dp.data_ptr(),
            rtot.data_ptr(), B, N, C, H, dx.data_ptr(), du_part.data_ptr(), dc_part.data_ptr(), dln_part.data_ptr(), st),
            "combo_bifuse_backward2_f32")
        # [B*chunks, 2C] -> [2C] by csrc/colsum.hip (ATen would split these sums over workgroups behind a memset node, which a
        # replayed hipGraph does not execute reliably: ops/colsum.py)
        from .colsum import channel_sum
        dgb = channel_sum(dgb_part, B * chunks, 2 * C, 1).view(2, C)
        dln = channel_sum(dln_part, B * chunks, 2 * C, 1).view(2, C)
        # inputs: x, ln_w, ln_b, eps, pos, u, c, z, b_ov, gamma_v, p_drop, drop_v, drop_a, seed
        return (dx, dln[0], dln[1], None, None, du_part.sum(1), dc_part.sum(1), dz_part.sum(1), dgb[1], dgb[0], None,
                None, None, None)


def token_op(x, ln_w, ln_b, eps, pos, u, c, z, b_ov, gamma_v, p_drop, drop_v=None, drop_a=None, seed=None):
    """x [B,N,256] token-major (level_embed already added) -> (y [B,N,256], pooled [B,8,256], spa [B,8]).
    pos [1,N,256] or [N,256].  p_drop > 0 uses the in-kernel Philox stream (fresh seed per call unless given);
    drop_v/drop_a inject explicit multipliers [B,8,N] instead (tests)."""
    if seed is None:
        _seed_counter[0] += 1
        seed = (0x9E3779B97F4A7C15 * _seed_counter[0] + 0x1234567) & 0xFFFFFFFFFFFFFFFF
    x = x.contiguous().float()
    pos = pos.reshape(-1, pos.shape[-1]).contiguous().float()
    return _BifuseTokenOp.apply(x, ln_w.float().contiguous(), ln_b.float().contiguous(), float(eps), pos,
                                u.contiguous().float(), c.contiguous().float(), z.contiguous().float(),
                                b_ov.float().contiguous(), gamma_v.float().contiguous(), float(p_drop), drop_v, drop_a,
                                int(seed))
