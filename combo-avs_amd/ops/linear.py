"""Dense layers on the bf16 matrix cores with fp32 accuracy (3-way bf16 split, csrc/gemm_x3.hip).
`linear(x, weight, bias, relu)` == F.linear (+ReLU) for fp32 CUDA tensors; forward, dX and dW all run on the same
HIP kernel (k-contiguous / row-contiguous operand loaders, split-K for dW)."""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib

MIN_ROWS = 512  # below this the op is launch/weight-bandwidth bound and the library GEMV path is as good
# Which GEMM serves the head's nn.Linear layers: "library" = hipBLASLt fp32 through torch (plain library GEMM),
# "x3" = csrc/gemm_x3.hip.  Measured on MI355X (tools/bench_gemm.py, 41160x256x1024): x3 forward 151 us vs 257 us,
# but the whole training step does not get faster yet (dX needs a transposed weight copy, dW is library either way),
# so "library" stays the default until the x3 kernel has a deeper load pipeline.
#   "library3x" = hipBLASLt with torch's allow_tf32 switch: gfx950 has no TF32/xf32 matrix instruction, and
#   hipBLASLt serves that mode with a 3-way bf16 split as well (measured error 4.4e-6 vs 2.9e-7 for plain fp32, i.e.
#   the same class as csrc/gemm_x3.hip; tools/blas_test.py).  Forward and dX run 2.2x faster in that mode, the
#   TN-layout dW GEMM is slower (809 vs 491 us), so dW stays in plain fp32 mode.
_IMPL = __import__("os").environ.get("COMBO_LINEAR_IMPL", "library3x")  # "library": plain fp32 library GEMMs (parity experiments)


def set_impl(name):
    global _IMPL
    assert name in ("library", "library3x", "x3")
    _IMPL = name


def gemm_tn_x3(dy, x, with_bias_grad=False, out=None, db_out=None):
    """dW[N,K] = dy[M,N]^T @ x[M,K] on csrc/gemm_tn.hip (fp32-accurate bf16x3 MFMA, split-K over the tokens);
    with_bias_grad: also return db[N] = dy.sum(0), accumulated in the same pass.  The split-K partials (and the
    bias partials) are finished by ONE reduce launch that writes straight into `out` / `db_out` when given
    (contiguous row blocks of a packed gradient, e.g. nn.MultiheadAttention's in_proj)."""
    lib = _lib.lib()
    M, N = dy.shape
    K = x.shape[1]
    splits = lib.combo_gemm_tn_splits(M, N, K)
    mchunk = (-(-M // splits) + 15) // 16 * 16
    splits = -(-M // mchunk)
    dev = dy.device
    part = torch.empty(splits, N, K, device=dev, dtype=torch.float32)
    dbp = torch.empty(splits, N, device=dev, dtype=torch.float32) if with_bias_grad else None
    st = _lib.current_stream()
    with _lib.timed("gemm_tn_x3", (M, N, K)):
        rc = lib.combo_gemm_tn_x3_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), part.data_ptr(), _lib.ptr(dbp),
                                      M, N, K, splits, st)
    _lib.check(rc, "combo_gemm_tn_x3_f32")
    if out is None and splits == 1 and (db_out is None):
        return (part[0], dbp[0]) if with_bias_grad else part[0]
    dw = out if out is not None else torch.empty(N, K, device=dev, dtype=torch.float32)
    db = None
    if with_bias_grad:
        db = db_out if db_out is not None else torch.empty(N, device=dev, dtype=torch.float32)
    if (N * K) % 4 == 0 and dw.is_contiguous() and dw.data_ptr() % 16 == 0 and (db is None or db.is_contiguous()):
        _lib.check(lib.combo_splitk_reduce_f32(part.data_ptr(), splits, N * K, dw.data_ptr(), _lib.ptr(dbp),
                                               N if with_bias_grad else 0, _lib.ptr(db), st), "combo_splitk_reduce_f32")
    else:
        torch.sum(part, 0, out=dw)
        if with_bias_grad:
            torch.sum(dbp, 0, out=db)
    return (dw, db) if with_bias_grad else dw


import os as _os
NT_MIN_ROWS = 16384 if _os.environ.get('COMBO_GEMM_NT', '1') == '1' else 1 << 60  # csrc/gemm_nt.hip needs enough 256-token tiles to fill the chip; below: hipBLASLt's 3xbf16 mode


NT_V2 = _os.environ.get('COMBO_GEMM_NT2', '1') == '1'  # csrc/gemm_nt2.hip (persistent, pre-split weights); 0: v1 (A/B)


def presplit(b):
    """bf16 hi/lo image of a 2-D fp32 view [N, K] (any strides: pass `w.t()` for W^T, no copy) for csrc/gemm_nt2.hip."""
    N, K = b.shape
    img = torch.empty(N, K, device=b.device, dtype=torch.float32)
    _lib.check(_lib.lib().combo_presplit_bf16x2_f32(b.data_ptr(), b.stride(0), b.stride(1), N, K, img.data_ptr(),
                                                    _lib.current_stream()), "combo_presplit_bf16x2_f32")
    return img


def gemm_nt_x3(a, b, bias=None, relu=False, relu_mask=None):
    """C[M,N] = a[M,K] @ b[N,K]^T (+ bias) (+ ReLU), fp32-accurate bf16x3 MFMA.  b may be any strided 2-D view (e.g.
    `weight.t()` for dX).  v2 (csrc/gemm_nt2.hip): weight pre-split once, persistent tiles; v1: csrc/gemm_nt.hip."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    if relu_mask is not None:  # C = relu_mask > 0 ? a @ b^T : 0 (the ReLU backward of the consumer, csrc/gemm_nt2.hip)
        assert bias is None and not relu and relu_mask.shape == (M, N) and relu_mask.is_contiguous()
        if NT_V2 and K % 16 == 0 and M * N * 4 < 2 ** 31 - 1:
            img = presplit(b)
            with _lib.timed("gemm_nt_x3", (M, N, K)):
                rc = _lib.lib().combo_gemm_nt_x3_pre_masked_f32(a.data_ptr(), a.stride(0), img.data_ptr(), relu_mask.data_ptr(),
                                                                out.data_ptr(), N, M, N, K, _lib.current_stream())
            _lib.check(rc, "combo_gemm_nt_x3_pre_masked_f32")
            return out
        return relu_grad(gemm_nt_x3(a, b), relu_mask)
    if NT_V2 and K % 16 == 0 and M * N * 4 < 2 ** 31 - 1:  # (v2 addresses C with 32-bit byte offsets)
        img = presplit(b)
        with _lib.timed("gemm_nt_x3", (M, N, K)):
            rc = _lib.lib().combo_gemm_nt_x3_pre_f32(a.data_ptr(), a.stride(0), img.data_ptr(), _lib.ptr(bias), out.data_ptr(),
                                                     N, M, N, K, 1 if relu else 0, _lib.current_stream())
        _lib.check(rc, "combo_gemm_nt_x3_pre_f32")
        return out
    if b.stride(1) != 1:
        b = b.contiguous()
    with _lib.timed("gemm_nt_x3", (M, N, K)):
        rc = _lib.lib().combo_gemm_nt_x3_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), _lib.ptr(bias),
                                             out.data_ptr(), N, M, N, K, 1 if relu else 0, _lib.current_stream())
    _lib.check(rc, "combo_gemm_nt_x3_f32")
    return out


# opt-in (COMBO_NT2_SMALL_MIN_ROWS=1024): the decoder's M = BT*100-token layers on gemm_nt2's skinny configuration.  Measured
# inside a captured graph (tools/sweep_nt.py, profiles/r01_gemm_routing_sweep.txt): 12.2 us incl. the weight pre-split
# against hipBLASLt's 10.4 us at 4000x256->256 - the library wins every shape below ~100 wide tiles.
NT2_SMALL_MIN_ROWS = int(_os.environ.get("COMBO_NT2_SMALL_MIN_ROWS", str(1 << 60)))


def _nt_ok(a, n_out, b=None):
    """Routing between the head's own forward / dX GEMM kernel and hipBLASLt's 3xbf16 mode, from the in-graph sweep
    (tools/sweep_nt.py): gemm_nt2 (+ its weight pre-split launch) wins once its 256 x 128 tiles number >= ~100 -
    25 vs 41 us at 31360x256->256, 26 vs 39 us at 4000x256->2048, 33 vs 78 us at 41160x256->288, 95 vs 123 us at
    16384x2048->256 - and loses on few tiles with a long K (94 vs 37 us at 4000x2048->256) and on the decoder's small
    layers (18 vs 10 us at 4000x256->256).  v1 (csrc/gemm_nt.hip, COMBO_GEMM_NT2=0) keeps its own, older thresholds."""
    tiles = -(-a.shape[0] // 256) * -(-n_out // 128)
    if NT_V2:
        big = a.shape[0] >= 2048 and tiles >= 120 and n_out >= 64 and NT_MIN_ROWS < (1 << 60)
    else:
        big = a.shape[0] >= NT_MIN_ROWS and tiles >= 256 and n_out >= 128
    small = NT_V2 and NT2_SMALL_MIN_ROWS <= a.shape[0] < NT_MIN_ROWS and tiles <= 128 and n_out >= 64
    ok = ((big or small) and a.shape[1] % 16 == 0 and a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0)
    if b is not None:
        ok = ok and b.stride(1) == 1 and b.stride(0) % 4 == 0 and b.data_ptr() % 16 == 0
    return ok


def relu_grad(dy, y):
    """dy * (y > 0) in one launch (csrc/biasact.hip)."""
    if dy.is_cuda and dy.dtype == torch.float32 and y.dtype == torch.float32 and dy.is_contiguous() and y.is_contiguous() \
            and dy.numel() % 4 == 0 and dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().combo_relu_grad_f32(dy.data_ptr(), y.data_ptr(), dy.numel(), dx.data_ptr(), _lib.current_stream()),
                   "combo_relu_grad_f32")
        return dx
    return dy * (y > 0)


class _split3:
    """context: route library GEMMs through hipBLASLt's bf16x3 path (torch spells the switch `allow_tf32`)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.prev = torch.backends.cuda.matmul.allow_tf32
        torch.backends.cuda.matmul.allow_tf32 = self.on

    def __exit__(self, *a):
        torch.backends.cuda.matmul.allow_tf32 = self.prev


class _LinearLib3x(Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias, relu, defer=False, mask_dx=False, grad_masked=False):
        """mask_dx: x2d is the ReLU output of the producing layer and feeds nothing else - the input gradient is returned
        already multiplied by [x2d > 0] (folded into the dX GEMM's epilogue); grad_masked (with relu): the consumer does
        exactly that, so the incoming gradient needs no ReLU-gradient pass.  Set in pairs by `ffn` below."""
        ctx.defer, ctx.mask_dx, ctx.grad_masked = defer, mask_dx, grad_masked
        if _nt_ok(x2d, weight.shape[0], weight) and (bias is None or bias.is_contiguous()):
            y = gemm_nt_x3(x2d, weight, bias, relu)  # bias + ReLU in the epilogue
        else:
            with _split3(True):
                y = torch.nn.functional.linear(x2d, weight, bias)
            if relu:
                y = torch.relu_(y)
        ctx.save_for_backward(x2d, weight, y if relu else None)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2d, weight, y = ctx.saved_tensors
        if ctx.relu and not ctx.grad_masked:
            dy = relu_grad(dy, y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dyc = dy if dy.stride(1) == 1 else dy.contiguous()
            if _nt_ok(dyc, weight.shape[1]):  # (W^T is read through a strided view by the weight pre-split: no transpose copy)
                dx = gemm_nt_x3(dyc, weight.t(), relu_mask=x2d if ctx.mask_dx else None)
            else:
                with _split3(True):
                    dx = dy @ weight
                if ctx.mask_dx:
                    dx = relu_grad(dx, x2d)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if ctx.defer and _dw_queue is not None:
                dyc = dy if dy.stride(1) == 1 else dy.contiguous()
                key = ("w", weight.data_ptr())
                ent = _dw_index.get(key)
                if ent is not None and _deferrable(dyc, x2d, ent[1]):
                    ent[0].append((dyc, x2d))  # another use of the same weight: joins the entry, autograd gets "no gradient"
                    return dx, None, None, None, None, None, None
                if ent is None:
                    dw_t = torch.empty_like(weight)
                    db_t = torch.empty(weight.shape[0], device=weight.device, dtype=weight.dtype) if want_db else None
                    if _deferrable(dyc, x2d, dw_t):
                        ent = [[(dyc, x2d)], dw_t, db_t]
                        _dw_queue.append(ent)  # computed by the grouped launch when deferred_dw() closes
                        _dw_index[key] = ent
                        return dx, dw_t, db_t, None, None, None, None
            if dy.stride(1) == 1 and x2d.stride(1) == 1 and dy.shape[0] >= 512:
                # long-reduction / tiny-output shape: 3x faster than the library GEMM; db rides along
                r = gemm_tn_x3(dy, x2d, with_bias_grad=want_db)
                dw, db = r if want_db else (r, None)
            else:
                with _split3(False):
                    dw = dy.t() @ x2d
        if want_db and db is None:
            db = dy.sum(0)
        return dx, dw, db, None, None, None, None


class _TnProblem(ctypes.Structure):  # combo_gemm_tn_problem (include/combo_avs.h)
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("partials", ctypes.c_void_p), ("db_partials", ctypes.c_void_p),
                ("ldy", ctypes.c_longlong), ("ldx", ctypes.c_longlong), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("splits", ctypes.c_int)]


class _RedProblem(ctypes.Structure):  # combo_reduce_problem
    _fields_ = [("partials", ctypes.c_void_p), ("out", ctypes.c_void_p), ("db_partials", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("n", ctypes.c_longlong), ("splits", ctypes.c_int), ("nb", ctypes.c_int)]


_GROUP_TOKENS_PER_SPLIT = int(_os.environ.get("COMBO_DW_TOKENS_PER_SPLIT", "1024"))
_dw_queue = None  # [[uses, dw_out, db_out]] while a deferred_dw() context is open (uses = [(dy, x2d), ...])
_dw_index = {}    # ("w" | "ln", parameter address) -> queue entry: repeated uses of one parameter join its entry
_ln_queue = None  # [(dy, x, mean, rstd, out[2,C])]: LayerNorm parameter gradients, same idea (ops/layernorm.py)


class _LnProblem(ctypes.Structure):  # combo_ln_grad_problem
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
                ("partials", ctypes.c_void_p), ("tokens", ctypes.c_longlong), ("C", ctypes.c_int), ("tokens_per_slice", ctypes.c_int)]


def _flush_ln(q):
    """q: [[uses, out[2,C]]] with uses = [(dy, x, mean, rstd), ...] (see _flush_dw for repeated uses)."""
    lib, st = _lib.lib(), _lib.current_stream()
    n_pr = sum(len(e[0]) for e in q)
    pr, red, keep = (_LnProblem * n_pr)(), (_RedProblem * len(q))(), []
    t, tps = 0, 64
    for i, (uses, out) in enumerate(q):
        C = out.shape[1]
        plan = [-(-u[0].shape[0] // tps) for u in uses]
        total = sum(plan)
        part = torch.empty(total, 2, C, device=out.device, dtype=torch.float32)
        keep.append(part)
        off = 0
        for (dy, x, mean, rstd), slices in zip(uses, plan):
            pr[t] = _LnProblem(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), part[off].data_ptr(), dy.shape[0], C, tps)
            t += 1
            off += slices
        red[i] = _RedProblem(part.data_ptr(), out.data_ptr(), 0, 0, 2 * C, total, 0)
    _lib.check(lib.combo_ln_param_grad_grouped_f32(ctypes.cast(pr, ctypes.c_void_p), n_pr, st), "combo_ln_param_grad_grouped_f32")
    _lib.check(lib.combo_splitk_reduce_grouped_f32(ctypes.cast(red, ctypes.c_void_p), len(q), st), "combo_splitk_reduce_grouped_f32")


class deferred_dw:
    """Inside this context the weight gradients of layers marked `defer=True` are not computed when autograd reaches them:
    (dY, X, destination) is queued and ONE grouped launch (+ one grouped reduce) computes them all when the context closes.
    The decoder's dW GEMMs are ~25 us of latency each for 0.5 GFLOP (114 per step) and nothing on the backward critical
    path reads them.  Only valid for weights that are used ONCE per forward (autograd would otherwise sum the not yet
    written tensors) and when the gradients are read after the context closes (trainer.FlatAdamW.backward does that)."""

    def __enter__(self):
        global _dw_queue, _ln_queue, _dw_index
        self.prev, _dw_queue = _dw_queue, []
        self.prev_ln, _ln_queue = _ln_queue, []
        self.prev_index, _dw_index = _dw_index, {}
        return self

    def __exit__(self, *exc):
        global _dw_queue, _ln_queue, _dw_index
        q, _dw_queue = _dw_queue, self.prev
        ql, _ln_queue = _ln_queue, self.prev_ln
        _dw_index = self.prev_index
        if exc[0] is None and q:
            _flush_dw(q)
        if exc[0] is None and ql:
            _flush_ln(ql)
        return False


def _deferrable(dy, x2d, dw_out):
    M, N = dy.shape
    K = x2d.shape[1]
    return (_dw_queue is not None and dy.stride(1) == 1 and x2d.stride(1) == 1 and M >= 256 and N % 4 == 0 and K % 4 == 0
            and N >= 64 and K >= 64 and dy.stride(0) % 4 == 0 and x2d.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0
            and x2d.data_ptr() % 16 == 0 and dw_out.is_contiguous() and dw_out.data_ptr() % 16 == 0 and (N * K) % 4 == 0)


def _flush_dw(q):
    """q: [[uses, dw_out, db_out]] with uses = [(dy, x2d), ...]: a weight that is applied several times per forward (the
    prediction heads run 10 times) is ONE entry - every use becomes its own GEMM problem writing its own slice of the
    entry's split-K partials, and one reduce sums all slices, i.e. the sum over the uses costs nothing extra."""
    lib, st = _lib.lib(), _lib.current_stream()
    n_tn = sum(len(e[0]) for e in q)
    tn, red, keep = (_TnProblem * n_tn)(), (_RedProblem * len(q))(), []
    t = 0
    for i, (uses, dw, db) in enumerate(q):
        N, K = dw.shape
        plan = []
        for dy, x2d in uses:
            M = dy.shape[0]
            # a problem of a grouped launch does not have to fill the chip alone: long token chunks per workgroup keep the
            # split-K partial traffic (and the reduce) small; the group as a whole still has thousands of workgroups
            splits = min(lib.combo_gemm_tn_splits(M, N, K), max(1, -(-M // _GROUP_TOKENS_PER_SPLIT)))
            mchunk = (-(-M // splits) + 15) // 16 * 16
            plan.append(-(-M // mchunk))
        total = sum(plan)
        part = torch.empty(total, N, K, device=dw.device, dtype=torch.float32)
        dbp = torch.empty(total, N, device=dw.device, dtype=torch.float32) if db is not None else None
        keep.append((part, dbp))
        off = 0
        for (dy, x2d), splits in zip(uses, plan):
            tn[t] = _TnProblem(dy.data_ptr(), x2d.data_ptr(), part[off].data_ptr(), dbp[off].data_ptr() if dbp is not None else 0,
                               dy.stride(0), x2d.stride(0), dy.shape[0], N, K, splits)
            t += 1
            off += splits
        red[i] = _RedProblem(part.data_ptr(), dw.data_ptr(), _lib.ptr(dbp), _lib.ptr(db), N * K, total, N if db is not None else 0)
    flops = sum(2.0 * dy.shape[0] * e[1].shape[0] * e[1].shape[1] for e in q for dy, _ in e[0])
    with _lib.timed("gemm_tn_x3_grouped", (flops, n_tn)):
        rc = lib.combo_gemm_tn_x3_grouped_f32(ctypes.cast(tn, ctypes.c_void_p), n_tn, st)
    _lib.check(rc, "combo_gemm_tn_x3_grouped_f32")
    _lib.check(lib.combo_splitk_reduce_grouped_f32(ctypes.cast(red, ctypes.c_void_p), len(q), st), "combo_splitk_reduce_grouped_f32")


def _dw_into(dy, x2d, dw_out, db_out, defer=False):
    """dW (+ db) of one projection, written into row blocks of a packed gradient."""
    if defer and _deferrable(dy, x2d, dw_out) and (db_out is None or db_out.is_contiguous()):
        _dw_queue.append([[(dy, x2d)], dw_out, db_out])
        return
    if dy.stride(1) == 1 and x2d.stride(1) == 1 and dy.shape[0] >= 512:
        gemm_tn_x3(dy, x2d, with_bias_grad=db_out is not None, out=dw_out, db_out=db_out)
    else:
        with _split3(False):
            torch.mm(dy.t(), x2d, out=dw_out)
        if db_out is not None:
            torch.sum(dy, 0, out=db_out)


class _InProj(Function):
    """q, k, v = nn.MultiheadAttention's packed input projection.  The packed [3E,E] weight is sliced INSIDE the node:
    sliced leaves would cost, per attention layer and step, 6 zero-filled [3E,E]/[3E] gradients + 6 slice copies +
    4 accumulation adds in autograd; here the three weight gradients land in row blocks of one [3E,E] tensor."""

    @staticmethod
    def forward(ctx, xq, xk, xv, W, b, same_qk, defer=False):
        ctx.defer = defer
        E = W.shape[1]
        with _split3(True):
            q = torch.nn.functional.linear(xq, W[:E], b[:E])
            k = torch.nn.functional.linear(xk, W[E:2 * E], b[E:2 * E])
            v = torch.nn.functional.linear(xv, W[2 * E:], b[2 * E:])
        ctx.save_for_backward(xq, xk, xv, W)
        ctx.same_qk = same_qk
        return q, k, v

    @staticmethod
    @once_differentiable
    def backward(ctx, dq, dk, dv):
        xq, xk, xv, W = ctx.saved_tensors
        E = W.shape[1]
        dq, dk, dv = dq.contiguous(), dk.contiguous(), dv.contiguous()
        dxq = dxk = dxv = None
        with _split3(True):
            if ctx.needs_input_grad[0]:
                dxq = dq @ W[:E]
                if ctx.same_qk:  # q and k read the same tensor: accumulate in the GEMM epilogue, not in autograd
                    dxq = torch.addmm(dxq, dk, W[E:2 * E])
            if ctx.needs_input_grad[1] and not ctx.same_qk:
                dxk = dk @ W[E:2 * E]
            if ctx.needs_input_grad[2]:
                dxv = dv @ W[2 * E:]
        dW = db = None
        if ctx.needs_input_grad[3]:
            dW = torch.empty_like(W)
            db = torch.empty(3 * E, device=W.device, dtype=W.dtype)
            for i, (dy, x) in enumerate(((dq, xq), (dk, xk), (dv, xv))):
                _dw_into(dy, x, dW[i * E:(i + 1) * E], db[i * E:(i + 1) * E], defer=ctx.defer)
        return dxq, dxk, dxv, dW, db, None, None


def in_proj(xq, xk, xv, weight, bias, same_qk=False, defer=False):
    """Packed q/k/v projection of nn.MultiheadAttention ([..., E] inputs -> three [..., E] outputs).  same_qk: xq and
    xk are the same tensor (self-attention); pass xk=None-equivalent semantics by giving the tensor twice."""
    E = weight.shape[1]
    if torch.is_autocast_enabled() or not xq.is_cuda or xq.dtype != torch.float32:
        return (linear(xq, weight[:E], bias[:E]), linear(xk, weight[E:2 * E], bias[E:2 * E]),
                linear(xv, weight[2 * E:], bias[2 * E:]))
    shp = (xq.shape[:-1], xk.shape[:-1], xv.shape[:-1])
    q, k, v = _InProj.apply(xq.reshape(-1, E), xk.reshape(-1, E), xv.reshape(-1, E), weight, bias, same_qk, defer)
    return q.view(*shp[0], E), k.view(*shp[1], E), v.view(*shp[2], E)


class _LinearCat(Function):
    """y = x @ cat(W1, W2)^T + cat(b1, b2): two nn.Linear layers that read the same input as ONE GEMM (forward, dX and dW
    each once); the weight gradients are returned as row blocks of the merged gradient."""

    @staticmethod
    def forward(ctx, x2d, w1, b1, w2, b2, defer=False):
        ctx.defer = defer
        W = torch.cat([w1, w2], 0)
        b = torch.cat([b1, b2], 0)
        if _nt_ok(x2d, W.shape[0], W):
            y = gemm_nt_x3(x2d, W, b)
        else:
            with _split3(True):
                y = torch.nn.functional.linear(x2d, W, b)
        ctx.save_for_backward(x2d, W)
        ctx.n1 = w1.shape[0]
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2d, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if _nt_ok(dy, W.shape[1]):
                dx = gemm_nt_x3(dy, W.t())
            else:
                with _split3(True):
                    dx = dy @ W
        dW = torch.empty_like(W)
        db = torch.empty(W.shape[0], device=W.device, dtype=W.dtype)
        _dw_into(dy, x2d, dW, db, defer=ctx.defer)
        n1 = ctx.n1
        return dx, dW[:n1], db[:n1], dW[n1:], db[n1:], None


def linear_cat(x, w1, b1, w2, b2, defer=False):
    """[..., K] -> [..., N1 + N2]; fp32 CUDA tensors."""
    K = x.shape[-1]
    y = _LinearCat.apply(x.reshape(-1, K), w1, b1, w2, b2, defer)
    return y.view(*x.shape[:-1], y.shape[-1])


def gemm_x3(A, a_rowc, B, b_rowc, M, N, K, bias=None, relu=False, splits=1):
    """C[M,N] = sum_k A(m,k) B(n,k); operands are 2-D contiguous fp32 tensors ([rows,K] or, if *_rowc, [K,rows])."""
    lib = _lib.lib()
    dev = A.device
    nz = lib.combo_gemm_x3_splits(K, splits)
    out = torch.empty((nz, M, N) if nz > 1 else (M, N), device=dev, dtype=torch.float32)
    rc = lib.combo_gemm_x3_f32(A.data_ptr(), A.shape[1], 1 if a_rowc else 0, B.data_ptr(), B.shape[1],
                               1 if b_rowc else 0, _lib.ptr(bias), out.data_ptr(), N, M, N, K, 1 if relu else 0, nz,
                               M * N, _lib.current_stream())
    if rc != 0:
        raise _lib.HipError(f"combo_gemm_x3_f32 failed with hipError_t {rc}: A{tuple(A.shape)} rowc={a_rowc} "
                            f"B{tuple(B.shape)} rowc={b_rowc} M={M} N={N} K={K} ptrs {A.data_ptr() % 16},{B.data_ptr() % 16}")
    return out.sum(0) if nz > 1 else out


class _LinearX3(Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias, relu):
        M, K = x2d.shape
        N = weight.shape[0]
        y = gemm_x3(x2d, False, weight, False, M, N, K, bias, relu)
        ctx.save_for_backward(x2d, weight, y if relu else None)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2d, weight, y = ctx.saved_tensors
        M, K = x2d.shape
        N = weight.shape[0]
        dy = dy.contiguous()
        if ctx.relu:
            dy = relu_grad(dy, y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX[M,K] = sum_n dY(m,n) W(n,k) = dY . (W^T)^T: with a transposed copy of the (small) weight both operands
            # are k-contiguous, the fast loader path (the row-contiguous loader works too but transposes through LDS)
            dx = gemm_x3(dy, False, weight.t().contiguous(), False, M, K, N)
        if ctx.needs_input_grad[1]:
            # dW[N,K] = dY^T X reduces over the M tokens: library GEMM (the row-contiguous x3 path is correct, see
            # tests/test_gemm_gpu.py, but not yet faster than hipBLASLt for this layout)
            dw = dy.t() @ x2d
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False, defer=False, mask_dx=False, grad_masked=False):
    """F.linear(x, weight, bias) [+ ReLU].  fp32 CUDA tensors with enough rows go to the bf16x3 MFMA kernel.
    mask_dx / grad_masked: see _LinearLib3x.forward (only honoured on that path; use `ffn`)."""
    K = x.shape[-1]
    N = weight.shape[0]
    rows = x.numel() // K
    if (_IMPL == "library3x" and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and rows >= MIN_ROWS and not torch.is_autocast_enabled()):
        y = _LinearLib3x.apply(x.reshape(rows, K), weight, bias, relu, defer, mask_dx, grad_masked)
        return y.view(*x.shape[:-1], N)
    if (_IMPL == "x3" and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and rows >= MIN_ROWS and K % 4 == 0
            and N % 4 == 0 and rows % 4 == 0 and not torch.is_autocast_enabled()
            and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0 and weight.is_contiguous()):
        y = _LinearX3.apply(x.reshape(rows, K).contiguous(), weight.contiguous(), bias, relu)
        return y.view(*x.shape[:-1], N)
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.relu(y) if relu else y


def ffn(x, w1, b1, w2, b2, defer=True):
    """linear2(relu(linear1(x))) of the transformer FFN blocks (msdeformattn.py:125-134, transformer_decoder.py:178-182).
    On the HIP GEMM path the ReLU backward is folded into the second layer's input-gradient GEMM (its epilogue multiplies
    dH = dY . W2 by [H > 0]): no ReLU-gradient pass over the 1024- / 2048-wide hidden tensor (read dH + H, write dH)."""
    K = x.shape[-1]
    rows = x.numel() // K
    fused = (_IMPL == "library3x" and x.is_cuda and x.dtype == torch.float32 and w1.dtype == torch.float32
             and rows >= MIN_ROWS and not torch.is_autocast_enabled() and FFN_FUSED_RELU_GRAD)
    h = linear(x, w1, b1, relu=True, defer=defer, grad_masked=fused)
    return linear(h, w2, b2, defer=defer, mask_dx=fused)


FFN_FUSED_RELU_GRAD = _os.environ.get("COMBO_FFN_FUSED_RELU_GRAD", "1") == "1"  # 0: separate ReLU-gradient kernel (A/B)


class Linear(torch.nn.Linear):
    """nn.Linear whose forward/backward GEMMs run on csrc/gemm_x3.hip (same parameters / state-dict names)."""

    defer_dw = False  # set by modules whose weights are used once per forward (see deferred_dw)

    def forward(self, x):
        return linear(x, self.weight, self.bias, defer=self.defer_dw)
