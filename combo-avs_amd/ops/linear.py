"""Dense layers of the head on its own MFMA kernels - one route, no library GEMM on the hot path:

  forward   y  = x W^T + b (+ReLU)   csrc/gemm_f32.hip   exact fp32 (v_mfma_f32_32x32x2_f32): forward values end in the
                                                          decoder's `sigmoid(logit) < 0.5` masks, where the 2^-17 error of
                                                          a bf16 split flips near-zero cells
  dX        = dy W                   csrc/gemm_nt3.hip   3-product bf16 split on the bf16 matrix cores (W^T pre-split from
                                                          a strided view, no transpose copy; ReLU backward in the epilogue)
  dW, db    = dy^T x, sum dy         csrc/gemm_tn.hip    3-product split, split-K over the tokens; inside `deferred_dw()`
                                                          all weight gradients of a step run as ONE grouped launch

`linear(x, weight, bias, relu)` == F.linear (+ReLU) for fp32 CUDA tensors (nn.Linear of msdeformattn.py:119-134,
ops/modules/ms_deform_attn.py:102-108,128, transformer_decoder.py:99-118, 50-58, 178-182, 216-219).  Shapes the kernels do
not take (K not a multiple of 16, misaligned views; the class head's 3-wide gradient) go through torch - a handful of tiny
launches per step, counted in DESIGN section 5.
"""
import ctypes
import os
import weakref
import os as _os

import math

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


# ------------------------------------------------------------------------------------------------- kernels
def _aligned_rows(t):
    return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def f32_ok(a, w):
    """operands of csrc/gemm_f32.hip: fp32, K-contiguous rows, 16-byte aligned, K % 16 == 0, 32-bit addressable output"""
    return (a.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32 and a.dim() == 2 and w.dim() == 2
            and a.shape[1] % 16 == 0 and a.shape[0] > 0 and _aligned_rows(a) and _aligned_rows(w)
            and a.shape[0] * w.shape[0] * 4 < 2 ** 31 - 1)


SPLITK_F32 = True  # K slices + finishing sum for long reductions with few output tiles (tools/bench_f32.py --no-splitk: A/B)


def gemm_nt_f32(a, w, bias=None, relu=False, out=None):
    """C[M,N] = a[M,K] @ w[N,K]^T (+ bias) (+ ReLU), exact fp32 on the matrix cores.  `out` may be a column block of a wider
    row-major matrix (stride(1) == 1)."""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    lib = _lib.lib()
    splits = lib.combo_gemm_nt_splitk_plan(M, N, K) if (SPLITK_F32 and K >= 1024 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0) else 1
    if splits > 1:  # a long reduction with few output tiles: K slices as the batch entries of one launch + a finishing sum
        ws = torch.empty(splits, M, N, device=a.device, dtype=torch.float32)
        with _lib.timed("gemm_nt_f32", (M, N, K)):
            rc = lib.combo_gemm_nt_splitk_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(bias), out.data_ptr(),
                                              out.stride(0), M, N, K, 1 if relu else 0, splits, ws.data_ptr(), _lib.current_stream())
        _lib.check(rc, "combo_gemm_nt_splitk_f32")
        return out
    with _lib.timed("gemm_nt_f32", (M, N, K)):
        rc = lib.combo_gemm_nt_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(bias), out.data_ptr(),
                                   out.stride(0), M, N, K, 1 if relu else 0, _lib.current_stream())
    _lib.check(rc, "combo_gemm_nt_f32")
    return out


class split_pieces:
    """`with split_pieces(f16):` the pre-split launches inside write fp16 hi / lo pieces (f16 true: the image of a FORWARD weight
    in the "f16x3" mode) instead of bf16 ones (csrc/gemm_x3.hip combo_presplit_pieces; host-side state of the library)."""

    def __init__(self, f16):
        self.f16 = 1 if f16 else 0

    def __enter__(self):
        self.prev = _lib.lib().combo_presplit_pieces(self.f16)

    def __exit__(self, *exc):
        _lib.lib().combo_presplit_pieces(self.prev)
        return False


def presplit(b, f16=False):
    """hi/lo image of a 2-D fp32 view [N, K] (any strides: pass `w.t()` for W^T, no copy) for csrc/gemm_nt3.hip: bf16 pieces
    (every gradient GEMM, the "x3" forward mode) or, f16, fp16 pieces (the "f16x3" forward mode)."""
    N, K = b.shape
    img = torch.empty(N, K, device=b.device, dtype=torch.float32)
    with split_pieces(f16):
        _lib.check(_lib.lib().combo_presplit_bf16x2_f32(b.data_ptr(), b.stride(0), b.stride(1), N, K, img.data_ptr(),
                                                        _lib.current_stream()), "combo_presplit_bf16x2_f32")
    return img


class _SplitProblem(ctypes.Structure):  # combo_presplit_problem (include/combo_avs.h)
    _fields_ = [("src", ctypes.c_void_p), ("img", ctypes.c_void_p), ("ld_row", ctypes.c_longlong), ("ld_col", ctypes.c_longlong),
                ("img_ld", ctypes.c_longlong), ("N", ctypes.c_int), ("K", ctypes.c_int), ("taps", ctypes.c_int), ("flip", ctypes.c_int)]


_split_images = None  # key -> image, while a grouped_presplit() context is open: weights announced by the forward pass
_split_pending = []   # [(b view [N,K], image view [N,K] (row pitch = the image's))] not yet split


def _split_key(b):
    return (b.data_ptr(), tuple(b.shape), tuple(b.stride()))


def expect_input_grad(*weights):
    """Forward-pass announcement: the backward pass will run dX = dY . cat(weights, 0) (weights [N_i, K] as stored).  Inside
    grouped_presplit() the pre-split images of ALL announced weights are made by one grouped launch when the first of them is
    needed (csrc/gemm_x3.hip presplit_grouped_kernel) instead of one launch per weight and step (148 in the S4 step)."""
    if _split_images is None or not weights[0].is_cuda or any(w.dtype != torch.float32 or w.shape[0] % 8 for w in weights):
        return
    key = tuple(_split_key(w) for w in weights)
    if key in _split_images:
        return
    K, n_tot = weights[0].shape[1], sum(w.shape[0] for w in weights)
    img = torch.empty(K, n_tot, device=weights[0].device, dtype=torch.float32)  # operand b = cat(weights).t(): [K, n_tot]
    _split_images[key] = img
    off = 0
    for w in weights:
        _split_pending.append((w.t(), img[:, off:off + w.shape[0]]))
        off += w.shape[0]


class grouped_presplit:
    """Spans the forward AND backward pass of one training step (trainer.train_step / GraphedTrainStep): forward nodes announce
    the weights whose input-gradient GEMMs will run (expect_input_grad), the first of those GEMMs splits all of them at once."""

    def __enter__(self):
        global _split_images, _fwd_images, _fwd_plan_next, _fwd_plan_flushed
        self.prev, _split_images = (_split_images, _split_pending[:]), {}
        self.prev_fwd, _fwd_images = (_fwd_images, _fwd_plan_next, _fwd_plan_flushed), {}
        _fwd_plan_next, _fwd_plan_flushed = {}, False
        del _split_pending[:]
        return self

    def __exit__(self, *exc):
        global _split_images, _fwd_images, _fwd_plan, _fwd_plan_next, _fwd_plan_flushed
        _split_images = self.prev[0]
        _split_pending[:] = self.prev[1]
        if exc[0] is None and self.prev_fwd[0] is None:  # the outermost context of a completed step: what it used is the next step's plan
            _fwd_plan = _fwd_plan_next
        _fwd_images, _fwd_plan_next, _fwd_plan_flushed = self.prev_fwd
        return False


def _expected_image(*weights):
    """-> the image announced by expect_input_grad(*weights) (splitting everything still pending first), or None"""
    if _split_images is None:
        return None
    img = _split_images.get(tuple(_split_key(w) for w in weights))
    if img is not None and _split_pending:
        pr = (_SplitProblem * len(_split_pending))()
        for i, (b, dst) in enumerate(_split_pending):
            pr[i] = _SplitProblem(b.data_ptr(), dst.data_ptr(), b.stride(0), b.stride(1), dst.stride(0), b.shape[0], b.shape[1])
        _lib.check(_lib.lib().combo_presplit_bf16x2_grouped_f32(ctypes.cast(pr, ctypes.c_void_p), len(_split_pending),
                                                                _lib.current_stream()), "combo_presplit_bf16x2_grouped_f32")
        del _split_pending[:]
    return img


def x3_ok(a, n_out):
    """operands of csrc/gemm_nt3.hip (A rows; the B image is made by `presplit`)"""
    return (a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.shape[1] % 16 == 0 and a.shape[0] > 0
            and _aligned_rows(a) and a.shape[0] * n_out * 4 < 2 ** 31 - 1)


def gemm_nt_x3(a, b, bias=None, relu=False, relu_mask=None, img=None, add=None):
    """C[M,N] = a[M,K] @ b[N,K]^T (+ bias) (+ add) (+ ReLU) with the 3-product bf16 split (~2^-17 relative per product): the
    input-gradient GEMM dX = dY . W (b = `weight.t()`, any strided 2-D view).  relu_mask [M,N]: C = relu_mask > 0 ? C : 0
    (the ReLU backward of the layer that produced the operand, folded into the epilogue); add [M,N]: another gradient arriving
    at the same tensor, summed in the epilogue (before the mask)."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    if img is None:
        img = presplit(b)
    lib, st = _lib.lib(), _lib.current_stream()
    for t in (relu_mask, add):
        assert t is None or (t.shape == (M, N) and t.is_contiguous() and t.dtype == torch.float32)
    splits = lib.combo_gemm_nt_x3_splitk_plan(M, N, K)
    ws = torch.empty(splits, M, N, device=a.device, dtype=torch.float32) if splits > 1 else None
    with _lib.timed("gemm_nt_x3", (M, N, K)):
        rc = lib.combo_gemm_nt_x3_epi2_f32(a.data_ptr(), a.stride(0), img.data_ptr(), _lib.ptr(bias), _lib.ptr(add), _lib.ptr(relu_mask),
                                           out.data_ptr(), N, M, N, K, 1 if relu else 0, splits, _lib.ptr(ws), st)
    _lib.check(rc, "combo_gemm_nt_x3_epi2_f32")
    return out


def gemm_tn_x3(dy, x, with_bias_grad=False, out=None, db_out=None):
    """dW[N,K] = dy[M,N]^T @ x[M,K] on csrc/gemm_tn.hip (3-product bf16 split, split-K over the tokens);
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


def relu_grad(dy, y):
    """dy * (y > 0) in one launch (csrc/biasact.hip)."""
    if dy.is_cuda and dy.dtype == torch.float32 and y.dtype == torch.float32 and dy.is_contiguous() and y.is_contiguous() \
            and dy.numel() % 4 == 0 and dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().combo_relu_grad_f32(dy.data_ptr(), y.data_ptr(), dy.numel(), dx.data_ptr(), _lib.current_stream()),
                   "combo_relu_grad_f32")
        return dx
    return dy * (y > 0)


def gemm_smallm_f32(a, w, bias=None, relu=False):
    """a[M <= 64, K] @ w[N, K]^T (+ bias) (+ ReLU): the weight-streaming kernel for audio_mlp (csrc/gemm_smallm.hip)"""
    M, K = a.shape
    N = w.shape[0]
    lib = _lib.lib()
    splits = lib.combo_gemm_smallm_splits(M, N, K)
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    part = torch.empty(splits, M, N, device=a.device, dtype=torch.float32) if splits > 1 else None
    with _lib.timed("gemm_smallm_f32", (M, N, K)):
        rc = lib.combo_gemm_smallm_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(bias), out.data_ptr(), N,
                                       _lib.ptr(part), splits, M, N, K, 1 if relu else 0, _lib.current_stream())
    _lib.check(rc, "combo_gemm_smallm_f32")
    return out


# ---- the head's bf16 throughput mode ---------------------------------------------------------------------------------------
# FORWARD_PRECISION = "bf16": every forward GEMM of the head (linear layers, 1x1 / 3x3 convolutions, the mask-logit
# contraction) runs on csrc/gemm_nt3.hip with ONE bf16 product per multiply-add (bf16 inputs rounded to nearest even, fp32
# accumulation): ~1/16 of the matrix-pipe time of the exact-fp32 instruction.  It is NOT the default: bf16 products move a
# mask logit by ~3e-3 of its scale, the north-star's 1e-3 bound needs the fp32 path.  Stated tolerance
# and its test: tests/test_head_gpu.py::test_bf16_forward_mode_stated_tolerance.  Gradient GEMMs keep the 3-product split.
DEFAULT_FORWARD_PRECISION = os.environ.get("COMBO_HEAD_FORWARD", "f16x3")  # (round 6: was "fp32"; see "f16x3" below)
FORWARD_PRECISION = DEFAULT_FORWARD_PRECISION
_fwd_images = None  # {weight view key: bf16 hi/lo image [N, K]} of the current step (grouped_presplit context)


# "f16x3" (round 6): the 3-product split on fp16 pieces - 22 mantissa bits per operand instead of bf16's 16, i.e. fp32-GRADE products
# (error ~2^-22 |x.w| per term against 2^-16 for "x3") at the same matrix-pipe cost; needs |x| < 65 504: forward activations and
# weights of the normalised head, never gradients.
FORWARD_MODES = ("fp32", "f16x3", "x3", "bf16")  # from exact to cheap
if DEFAULT_FORWARD_PRECISION not in FORWARD_MODES:
    raise ValueError(f"COMBO_HEAD_FORWARD={DEFAULT_FORWARD_PRECISION!r}: one of {FORWARD_MODES} expected")


def set_forward_precision(mode):
    global FORWARD_PRECISION
    if mode not in FORWARD_MODES:
        raise ValueError(mode)
    FORWARD_PRECISION = mode


class forward_precision_scope:
    """`with forward_precision_scope("x3"):` - the forward GEMMs issued inside run in `mode` unless the global mode is already a
    cheaper one ("bf16" stays "bf16"); restores the previous mode on exit.  Used by the pixel decoder (modeling/pixel_decoder.py
    PIXEL_DECODER_FORWARD): its forward GEMMs sit in FRONT of the first thresholded attention mask, like the backbones'."""
    _rank = {m: i for i, m in enumerate(FORWARD_MODES)}

    def __init__(self, mode):
        if mode not in self._rank:
            raise ValueError(mode)
        self.mode = mode

    _base = None  # the mode in force outside the outermost open scope (nested scopes are judged against IT, not against each other)

    def __enter__(self):
        global FORWARD_PRECISION
        cls = forward_precision_scope
        self.prev, self.outer = FORWARD_PRECISION, cls._base is None
        if self.outer:
            cls._base = FORWARD_PRECISION
        FORWARD_PRECISION = self.mode if self._rank[self.mode] > self._rank[cls._base] else cls._base
        return self

    def __exit__(self, *exc):
        global FORWARD_PRECISION
        FORWARD_PRECISION = self.prev
        if self.outer:
            forward_precision_scope._base = None
        return False


class range_safe:
    """`with range_safe():` - forward GEMMs / convolutions whose operand is NOT a normalised activation (the pixel decoder's input
    projections and lateral convolutions read the backbones' ReLU features as they come): the fp16 pieces of "f16x3" end at 65 504
    (hi = inf, lo = x - inf: NaN), so inside the block that mode gives way to the exact fp32 instruction; the other modes (bf16
    pieces share fp32's range) stay.  tools/probe_f16_range.py: 40 steps at 20 x the learning rate drive res5 to 6.6e4."""

    def __enter__(self):
        global FORWARD_PRECISION
        self.prev = FORWARD_PRECISION
        if FORWARD_PRECISION == "f16x3":
            FORWARD_PRECISION = "fp32"
        return self

    def __exit__(self, *exc):
        global FORWARD_PRECISION
        FORWARD_PRECISION = self.prev
        return False


def forward_products():
    """`products` argument of the head's forward GEMMs in the current mode (combo_gemm_nt2_products): "bf16": 1; "x3": the 3-product
    split on bf16 pieces; "f16x3": 19 = the same on fp16 pieces"""
    return 1 if FORWARD_PRECISION == "bf16" else 19 if FORWARD_PRECISION == "f16x3" else 3


def forward_f16():
    return FORWARD_PRECISION == "f16x3"


# Forward images of a step, grouped (round 6).  With the forward GEMMs on the 3-product kernel every forward weight needs its hi / lo
# image once per step: 117 launches of ~6 us in the S4 step (0.7 ms).  The set of weights is the same every step, so a step REMEMBERS
# the weights it split (`_fwd_plan`: parameters and views of parameters only - their memory is final when the step starts; a tensor
# computed during the step, e.g. a concatenated or re-laid-out weight, could be read before it is written) and the next step splits all
# of them with ONE grouped launch per piece type when the first one is asked for.  Only the weight views are kept from step to step;
# the images are allocated per step (inside a captured graph: from the graph's pool), so a replayed graph never depends on a buffer
# the plan owns.  A planned weight that a step does not use costs its split and leaves the plan at the end of that step.
FORWARD_PLAN = True  # (tools/ab_const.py flips it for the A/B)
_fwd_plan = {}        # (view key, f16) -> weight view: the forward weights of the previous completed step
_fwd_plan_next = {}   # ... of the step in progress
_fwd_plan_flushed = False


def reset_forward_plan():
    """forget the forward weights of the previous step (the plan holds views of the parameters: call it when a model is dropped)"""
    global _fwd_plan
    _fwd_plan = {}


def _plannable(w):
    base = w._base if w._base is not None else w
    return isinstance(base, torch.nn.Parameter) and w.dim() == 2 and w.dtype == torch.float32 and w.shape[1] % 8 == 0


def _flush_forward_plan():
    """one grouped pre-split launch per piece type for every weight of the plan -> _fwd_images"""
    global _fwd_plan_flushed
    _fwd_plan_flushed = True
    dev = torch.cuda.current_device()
    for f16 in (False, True):
        todo = [(k, w) for k, w in _fwd_plan.items()
                if k[1] == f16 and k not in _fwd_images and k[0] == _split_key(w) and w.is_cuda and w.device.index == dev]
        if not todo:
            continue
        buf = torch.empty(sum(w.shape[0] * w.shape[1] for _, w in todo), device=todo[0][1].device, dtype=torch.float32)
        pr, off = (_SplitProblem * len(todo))(), 0
        for i, (k, w) in enumerate(todo):
            N, K = w.shape
            img = buf[off:off + N * K].view(N, K)
            off += N * K
            pr[i] = _SplitProblem(w.data_ptr(), img.data_ptr(), w.stride(0), w.stride(1), K, N, K)
            _fwd_images[k] = img
            _fwd_plan_next[k] = w
        with split_pieces(f16):
            _lib.check(_lib.lib().combo_presplit_bf16x2_grouped_f32(ctypes.cast(pr, ctypes.c_void_p), len(todo), _lib.current_stream()),
                       "combo_presplit_bf16x2_grouped_f32 (forward images)")


def forward_image(weight):
    """hi/lo image of a forward weight [N, K] in the current mode's piece type (cached for the step inside grouped_presplit())"""
    f16 = forward_f16()
    if _fwd_images is None:
        return presplit(weight, f16)
    key = (_split_key(weight), f16)
    img = _fwd_images.get(key)
    if img is None and FORWARD_PLAN and not _fwd_plan_flushed and key in _fwd_plan and weight.is_cuda:
        _flush_forward_plan()
        img = _fwd_images.get(key)
    if img is None:
        img = _fwd_images[key] = presplit(weight, f16)
        if FORWARD_PLAN and _plannable(weight):
            _fwd_plan_next[key] = weight.detach()
    return img


def gemm_nt_bf16(a, w, bias=None, relu=False, out=None, img=None):
    """C = a @ w^T (+ bias) (+ ReLU) with ONE bf16 product per multiply-add, fp32 accumulation (csrc/gemm_nt3.hip)"""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    if img is None:
        img = forward_image(w)
    lib, st = _lib.lib(), _lib.current_stream()
    # a long reduction with few output tiles (the decoder FFN's linear2: 4000 x 2048 -> 256; the res5 / res4 input projections): K slices
    # as the batch entries of one launch + the fixed-order finishing sum, as the exact path and the gradient GEMMs do
    splits = lib.combo_gemm_nt_x3_splitk_plan(M, N, K) if (FORWARD_SPLITK and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0
                                                           and (bias is None or bias.data_ptr() % 16 == 0)) else 1
    ws = torch.empty(splits, M, N, device=a.device, dtype=torch.float32) if splits > 1 else None
    prev = lib.combo_gemm_nt2_products(forward_products())
    try:
        with _lib.timed("gemm_nt_bf16", (M, N, K)):
            if splits > 1:
                rc = lib.combo_gemm_nt_x3_pre_splitk_f32(a.data_ptr(), a.stride(0), img.data_ptr(), _lib.ptr(bias), out.data_ptr(), out.stride(0),
                                                         M, N, K, 1 if relu else 0, splits, ws.data_ptr(), st)
            else:
                rc = lib.combo_gemm_nt_x3_pre_f32(a.data_ptr(), a.stride(0), img.data_ptr(), _lib.ptr(bias), out.data_ptr(), out.stride(0),
                                                  M, N, K, 1 if relu else 0, st)
    finally:
        lib.combo_gemm_nt2_products(prev)
    _lib.check(rc, "combo_gemm_nt_x3_pre_f32 (3-product / bf16 forward mode)")
    return out


FORWARD_SPLITK = True  # (tools/ab_const.py flips it for the A/B)


def _bf16_ok(x2d, weight, out):
    return (FORWARD_PRECISION != "fp32" and x3_ok(x2d, weight.shape[0]) and weight.dtype == torch.float32 and weight.dim() == 2
            and weight.shape[1] % 16 == 0 and weight.shape[0] % 4 == 0 and x2d.shape[0] > 64
            and (out is None or (out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0)))


def forward_gemm(x2d, weight, bias, relu, out=None):
    if _bf16_ok(x2d, weight, out) and (bias is None or bias.is_contiguous()):
        return gemm_nt_bf16(x2d, weight, bias, relu, out)
    if f32_ok(x2d, weight) and (bias is None or bias.is_contiguous()):
        if out is None and x2d.shape[0] <= 64 and x2d.shape[1] % 64 == 0 and weight.shape[0] * x2d.shape[1] >= (1 << 18):
            return gemm_smallm_f32(x2d, weight, bias, relu)  # a few rows against a large weight: stream the weight
        return gemm_nt_f32(x2d, weight, bias, relu, out)
    y = torch.nn.functional.linear(x2d, weight, bias)  # shapes outside the kernel's contract (see the module docstring)
    if relu:
        y = torch.relu_(y)
    if out is not None:
        out.copy_(y)
        return out
    return y


def input_grad_gemm(dy, weight, relu_mask=None, add=None):
    """dX = dy @ weight (weight [N,K] as stored, or a tuple of weights standing for their row concatenation), optionally
    + add (another gradient arriving at the same tensor, summed in the GEMM's epilogue), optionally multiplied by
    [relu_mask > 0]"""
    ws = weight if isinstance(weight, tuple) else (weight,)
    dyc = dy if _aligned_rows(dy) else dy.contiguous()
    if x3_ok(dyc, ws[0].shape[1]) and ws[0].dtype == torch.float32:
        img = _expected_image(*ws)
        if img is not None:
            return gemm_nt_x3(dyc, img, relu_mask=relu_mask, img=img, add=add)  # img [K, N] has the shape of the operand view
        return gemm_nt_x3(dyc, (ws[0] if len(ws) == 1 else torch.cat(ws, 0)).t(), relu_mask=relu_mask, add=add)
    w = ws[0] if len(ws) == 1 else torch.cat(ws, 0)
    if (dy.is_cuda and dy.dtype == torch.float32 and w.dtype == torch.float32 and dy.dim() == 2 and dy.shape[1] <= 16 and dy.stride(1) == 1
            and w.shape[1] % 4 == 0 and w.is_contiguous() and w.data_ptr() % 16 == 0 and relu_mask is None and add is None):
        # a tiny reduction length (class_embed: K + 1 classes): outer-product kernel instead of a BLAS tile GEMM
        dx = torch.empty(dy.shape[0], w.shape[1], device=dy.device, dtype=torch.float32)
        _lib.check(_lib.lib().combo_gemm_smallk_f32(dy.data_ptr(), dy.stride(0), w.data_ptr(), w.stride(0), dx.data_ptr(), dx.stride(0),
                                                    dy.shape[0], w.shape[1], dy.shape[1], _lib.current_stream()), "combo_gemm_smallk_f32")
        return dx
    dx = dy @ w
    if add is not None:
        dx = dx + add
    return relu_grad(dx, relu_mask) if relu_mask is not None else dx


# ------------------------------------------------------------------------------ deferred, grouped weight gradients
class _TnProblem(ctypes.Structure):  # combo_gemm_tn_problem (include/combo_avs.h)
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("partials", ctypes.c_void_p), ("db_partials", ctypes.c_void_p),
                ("ldy", ctypes.c_longlong), ("ldx", ctypes.c_longlong), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("splits", ctypes.c_int)]


class _RedProblem(ctypes.Structure):  # combo_reduce_problem
    _fields_ = [("partials", ctypes.c_void_p), ("out", ctypes.c_void_p), ("db_partials", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("n", ctypes.c_longlong), ("splits", ctypes.c_int), ("nb", ctypes.c_int)]


class _LnProblem(ctypes.Structure):  # combo_ln_grad_problem
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
                ("partials", ctypes.c_void_p), ("tokens", ctypes.c_longlong), ("C", ctypes.c_int), ("tokens_per_slice", ctypes.c_int)]


_GROUP_TOKENS_PER_SPLIT = 1024  # token slice of a grouped weight-gradient problem (tools/bench_dw.py sweep)
_dw_queue = None  # [[uses, dw_out, db_out, extras]] while a deferred_dw() context is open (uses = [(dy, x2d), ...])
_dw_index = {}    # ("w" | "ln", parameter address) -> queue entry: repeated uses of one parameter join its entry
_ln_queue = None  # [[uses, out[2,C]]]: LayerNorm parameter gradients, same idea (ops/layernorm.py)
_after_flush = []  # callables run once the deferred gradients exist (e.g. the FrozenBN fold of the backbone weight gradients)


def is_deferred_dest(t):
    """t is (a view of) the not-yet-written destination of a deferred weight gradient"""
    return _dw_queue is not None and any(e[1].data_ptr() == t.data_ptr() for e in _dw_queue)


def after_flush(fn):
    """run fn() when the open deferred_dw() context has written its gradients (at once without a context)"""
    if _dw_queue is None:
        fn()
    else:
        _after_flush.append(fn)


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
    path reads them.  The FIRST use of a weight hands autograd the (not yet written) destination tensor, later uses join the
    entry and return no gradient; a use the grouped kernel cannot take is computed at once into a side buffer that is added
    after the grouped reduce.  The gradients must be read after the context closes (trainer.FlatAdamW.backward does that)."""

    def __enter__(self):
        global _dw_queue, _ln_queue, _dw_index, _packed_grads
        self.prev, _dw_queue = _dw_queue, []
        self.prev_ln, _ln_queue = _ln_queue, []
        self.prev_index, _dw_index = _dw_index, {}
        self.prev_packed, _packed_grads = _packed_grads, {}
        return self

    def __exit__(self, *exc):
        global _dw_queue, _ln_queue, _dw_index, _packed_grads
        q, _dw_queue = _dw_queue, self.prev
        ql, _ln_queue = _ln_queue, self.prev_ln
        _dw_index = self.prev_index
        if exc[0] is None and q:
            _flush_dw(q)
        if exc[0] is None and ql:
            _flush_ln(ql)
        post, _after_flush[:] = _after_flush[:], []
        if exc[0] is None:
            _finish_packed_grads()
        _packed_grads = self.prev_packed
        if exc[0] is None:
            for fn in post:
                fn()
        return False


def _dest_ok(dw_out):
    """destination of a grouped weight-gradient problem ([N, K] written by the grouped reduce)"""
    N, K = dw_out.shape
    return (N % 4 == 0 and K % 4 == 0 and N >= 64 and K >= 64 and dw_out.is_contiguous() and dw_out.data_ptr() % 16 == 0)


def _use_ok(dy, x2d):
    """operands of a grouped weight-gradient problem (csrc/gemm_tn.hip, LDS-DMA kernel)"""
    return (dy.stride(1) == 1 and x2d.stride(1) == 1 and dy.shape[0] >= 256 and dy.stride(0) % 4 == 0
            and x2d.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and x2d.data_ptr() % 16 == 0)


def _deferrable(dy, x2d, dw_out):
    return _dw_queue is not None and _dest_ok(dw_out) and _use_ok(dy, x2d)


def _flush_dw(q):
    """q: [[uses, dw_out, db_out, extras]] with uses = [(dy, x2d), ...]: a weight that is applied several times per forward
    (the prediction heads run 10 times) is ONE entry - every use becomes its own GEMM problem writing its own slice of the
    entry's split-K partials, and one reduce sums all slices, i.e. the sum over the uses costs nothing extra.  extras =
    [(dw, db)] of uses that were computed immediately (not deferrable): added once the reduce has written the destination."""
    lib, st = _lib.lib(), _lib.current_stream()
    for _uses, dw, db, _extras in q:
        if not _uses:  # every use of this weight was computed at once (rare): the destination starts from zero
            dw.zero_()
            if db is not None:
                db.zero_()
    full = [e for e in q if e[0]]
    n_tn = sum(len(e[0]) for e in full)
    tn, red, keep = (_TnProblem * max(n_tn, 1))(), (_RedProblem * max(len(full), 1))(), []
    t = 0
    for i, (uses, dw, db, _extras) in enumerate(full):
        N, K = dw.shape
        plan = []
        for dy, x2d in uses:
            M = dy.shape[0]
            # a problem of a grouped launch does not have to fill the chip alone: long token chunks per workgroup keep the
            # split-K partial traffic (and the reduce) small; the group as a whole still has thousands of workgroups
            # (very long token axes - 1.3 M tokens at 512 x 512 inputs - get proportionally longer chunks: at most ~128 partials)
            splits = min(lib.combo_gemm_tn_splits(M, N, K), max(1, -(-M // max(_GROUP_TOKENS_PER_SPLIT, M // 128))))
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
    if n_tn:
        flops = sum(2.0 * dy.shape[0] * e[1].shape[0] * e[1].shape[1] for e in full for dy, _ in e[0])
        with _lib.timed("gemm_tn_x3_grouped", (flops, n_tn)):
            rc = lib.combo_gemm_tn_x3_grouped_f32(ctypes.cast(tn, ctypes.c_void_p), n_tn, st)
        _lib.check(rc, "combo_gemm_tn_x3_grouped_f32")
        _lib.check(lib.combo_splitk_reduce_grouped_f32(ctypes.cast(red, ctypes.c_void_p), len(full), st),
                   "combo_splitk_reduce_grouped_f32")
    for _uses, dw, db, extras in q:
        for edw, edb in extras:
            dw.add_(edw)
            if db is not None and edb is not None:
                db.add_(edb)


def _dw_now(dy, x2d, want_db):
    """dW (+ db) of one use, computed at once"""
    if dy.stride(1) == 1 and x2d.stride(1) == 1 and dy.shape[0] >= 512 and dy.shape[1] % 4 == 0 and x2d.shape[1] % 4 == 0 \
            and dy.shape[1] >= 64 and x2d.shape[1] >= 64 and dy.stride(0) % 4 == 0 and x2d.stride(0) % 4 == 0 \
            and dy.data_ptr() % 16 == 0 and x2d.data_ptr() % 16 == 0:
        r = gemm_tn_x3(dy, x2d, with_bias_grad=want_db)  # long-reduction / small-output shape; db rides along
        return r if want_db else (r, None)
    if (dy.is_cuda and dy.dtype == torch.float32 and x2d.dtype == torch.float32 and dy.shape[1] <= 16 and dy.stride(1) == 1
            and x2d.stride(1) == 1 and x2d.shape[1] % 4 == 0 and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0
            and dy.shape[0] >= 256):
        # a tiny output-row count (class_embed: K + 1 classes): FMA kernel + the split-K reduce instead of a BLAS tile GEMM
        lib, st = _lib.lib(), _lib.current_stream()
        M, N, K = dy.shape[0], dy.shape[1], x2d.shape[1]
        slices = lib.combo_gemm_tn_smalln_slices(M)
        part = torch.empty(slices, N, K, device=dy.device, dtype=torch.float32)
        dbp = torch.empty(slices, N, device=dy.device, dtype=torch.float32) if want_db else None
        _lib.check(lib.combo_gemm_tn_smalln_f32(dy.data_ptr(), dy.stride(0), x2d.data_ptr(), x2d.stride(0), M, N, K, part.data_ptr(),
                                                _lib.ptr(dbp), st), "combo_gemm_tn_smalln_f32")
        dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
        db = torch.empty(N, device=dy.device, dtype=torch.float32) if want_db else None
        _lib.check(lib.combo_splitk_reduce_f32(part.data_ptr(), slices, N * K, dw.data_ptr(), _lib.ptr(dbp), N if want_db else 0,
                                               _lib.ptr(db), st), "combo_splitk_reduce_f32")
        return dw, db
    dw = dy.t() @ x2d
    return dw, (dy.sum(0) if want_db else None)


_grad_targets = None  # {parameter address: view of the optimiser's flat gradient buffer} while grad_targets() is open


class grad_targets:
    """trainer.FlatAdamW.backward: the weight gradients computed by this module's kernels are written STRAIGHT into the flat
    gradient buffer (the views registered here) instead of into fresh tensors that one 350 MB concatenation then copies there."""

    def __init__(self, mapping):
        self.mapping = mapping

    def __enter__(self):
        global _grad_targets
        self.prev, _grad_targets = _grad_targets, self.mapping

    def __exit__(self, *exc):
        global _grad_targets
        _grad_targets = self.prev


_grad_alias = {}  # {address of a derived weight (the FrozenBN-folded copy a convolution runs on): address of its parameter}


def register_grad_aliases(derived, params):
    """backbone._FoldAll.forward: `derived[i]` is computed from the parameter `params[i]` by a per-channel scale, and its
    gradient node hands the scaled gradient on in place - so the gradient of `derived[i]` may be written where the gradient of
    `params[i]` lives.  Every call replaces the previous generation of aliases of the same parameters (the derived tensors of
    an earlier forward pass are gone; a recycled address must not resolve to a flat-buffer view)."""
    fresh = {p.data_ptr() for p in params}
    for k in [k for k, v in _grad_alias.items() if v in fresh]:
        del _grad_alias[k]
    for d, p in zip(derived, params):
        _grad_alias[d.data_ptr()] = p.data_ptr()


def drop_grad_aliases(param_ptrs):
    """backbone._FoldAll.backward: the aliases of these parameters' derived weights are no longer needed"""
    dead = set(param_ptrs)
    for k in [k for k, v in _grad_alias.items() if v in dead]:
        del _grad_alias[k]


def grad_target(ptr, shape, dtype):
    """the registered flat-buffer view for the parameter (or aliased derived weight) at address `ptr`, viewed as `shape`; None
    when there is none (no grad_targets() open, an unknown address, another dtype / element count)"""
    if _grad_targets is None:
        return None
    v = _grad_targets.get(ptr)
    if v is None and ptr in _grad_alias:
        v = _grad_targets.get(_grad_alias[ptr])
    if v is None or v.dtype != dtype or v.numel() != math.prod(shape):
        return None
    return v if tuple(v.shape) == tuple(shape) else v.view(shape)


def grad_buffer_like(weight):
    """destination of a full weight gradient: the registered flat-buffer view of `weight`, else a fresh tensor"""
    v = grad_target(weight.data_ptr(), weight.shape, weight.dtype) if weight.is_contiguous() else None
    return v if v is not None else torch.empty_like(weight)


def weight_grad(weight, dy, x2d, want_db, defer):
    """-> (dw, db) to hand to autograd for this use of `weight` (None, None when the use joined a deferred entry).
    Inside deferred_dw() a weight whose gradient the grouped reduce can write is entry-managed from its FIRST use on: that
    use hands autograd the (not yet written) destination, every use either becomes a problem of the grouped launch or - when
    the grouped kernel cannot take its operands - is computed at once into a side buffer that is added after the reduce.
    autograd therefore never adds anything to the unwritten destination."""
    if defer and _dw_queue is not None:
        dyc = dy if dy.stride(1) == 1 else dy.contiguous()
        key = ("w", weight.data_ptr())
        ent = _dw_index.get(key)
        first = ent is None
        if first:
            dw_t = grad_buffer_like(weight)
            if _dest_ok(dw_t):
                db_t = torch.empty(weight.shape[0], device=weight.device, dtype=weight.dtype) if want_db else None
                ent = [[], dw_t, db_t, []]
                _dw_queue.append(ent)  # finished by the grouped launch + reduce when deferred_dw() closes
                _dw_index[key] = ent
        if ent is not None:
            if _use_ok(dyc, x2d):
                ent[0].append((dyc, x2d))
            else:
                ent[3].append(_dw_now(dyc, x2d, ent[2] is not None))
            return (ent[1], ent[2]) if first else (None, None)
    return _dw_now(dy, x2d, want_db)


def _dw_into(dy, x2d, dw_out, db_out, defer=False):
    """dW (+ db) of one projection, written into row blocks of a packed gradient (each block is written exactly once)."""
    if defer and _deferrable(dy, x2d, dw_out) and (db_out is None or db_out.is_contiguous()):
        _dw_queue.append([[(dy, x2d)], dw_out, db_out, []])
        return
    if dy.stride(1) == 1 and x2d.stride(1) == 1 and dy.shape[0] >= 512 and dy.stride(0) % 4 == 0 and x2d.stride(0) % 4 == 0 \
            and dy.data_ptr() % 16 == 0 and x2d.data_ptr() % 16 == 0 and dy.shape[1] % 4 == 0 and x2d.shape[1] % 4 == 0 \
            and dy.shape[1] >= 64 and x2d.shape[1] >= 64:
        gemm_tn_x3(dy, x2d, with_bias_grad=db_out is not None, out=dw_out, db_out=db_out)
    else:
        torch.mm(dy.t(), x2d, out=dw_out)
        if db_out is not None:
            torch.sum(dy, 0, out=db_out)


# ------------------------------------------------------------------------------------------------- autograd nodes
class _Linear(Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias, relu, defer=False, mask_dx=False, grad_masked=False):
        """mask_dx: x2d is the ReLU output of the producing layer and feeds nothing else - the input gradient is returned
        already multiplied by [x2d > 0] (folded into the dX GEMM's epilogue); grad_masked (with relu): the consumer does
        exactly that, so the incoming gradient needs no ReLU-gradient pass.  Set in pairs by `ffn` below."""
        ctx.defer, ctx.mask_dx, ctx.grad_masked = defer, mask_dx, grad_masked
        y = forward_gemm(x2d, weight, bias, relu)
        if ctx.needs_input_grad[0]:
            expect_input_grad(weight)
        ctx.save_for_backward(x2d, weight, y if relu else None)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2d, weight, y = ctx.saved_tensors
        if ctx.relu and not ctx.grad_masked:
            dy = relu_grad(dy.contiguous(), y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = input_grad_gemm(dy, weight, relu_mask=x2d if ctx.mask_dx else None)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = weight_grad(weight, dy, x2d, want_db, ctx.defer)
        elif want_db:
            db = dy.sum(0)
        return dx, dw, db, None, None, None, None


class _InProj(Function):
    """q, k, v = nn.MultiheadAttention's packed input projection (transformer_decoder.py:99-118, 50-58).  The packed [3E,E]
    weight is sliced INSIDE the node: sliced leaves would cost, per attention layer and step, 6 zero-filled [3E,E]/[3E]
    gradients + 6 slice copies + 4 accumulation adds in autograd; here the three weight gradients land in row blocks of one
    [3E,E] tensor.  Self-attention (same_qk): q and k come out of ONE GEMM against the first 2E weight rows."""

    @staticmethod
    def forward(ctx, xq, xk, xv, W, b, same_qk, defer=False):
        ctx.defer = defer
        E = W.shape[1]
        if same_qk:
            qk = forward_gemm(xq, W[:2 * E], b[:2 * E], False)
            q, k = qk[:, :E], qk[:, E:]
        else:
            q = forward_gemm(xq, W[:E], b[:E], False)
            k = forward_gemm(xk, W[E:2 * E], b[E:2 * E], False)
        v = forward_gemm(xv, W[2 * E:], b[2 * E:], False)
        if same_qk:
            if ctx.needs_input_grad[0]:
                expect_input_grad(W[:2 * E])
        else:
            if ctx.needs_input_grad[0]:
                expect_input_grad(W[:E])
            if ctx.needs_input_grad[1]:
                expect_input_grad(W[E:2 * E])
        if ctx.needs_input_grad[2]:
            expect_input_grad(W[2 * E:])
        ctx.save_for_backward(xq, xk, xv, W)
        ctx.same_qk = same_qk
        return q, k, v

    @staticmethod
    @once_differentiable
    def backward(ctx, dq, dk, dv):
        xq, xk, xv, W = ctx.saved_tensors
        E = W.shape[1]
        dxq = dxk = dxv = None
        if ctx.same_qk:
            # q and k read the same tensor: one input-gradient GEMM over the concatenated [dq | dk] (K = 2E)
            dqk = torch.cat([dq, dk], 1)
            dq, dk = dqk[:, :E], dqk[:, E:]
            if ctx.needs_input_grad[0]:
                dxq = input_grad_gemm(dqk, W[:2 * E])
        else:
            dq, dk = dq.contiguous(), dk.contiguous()
            if ctx.needs_input_grad[0]:
                dxq = input_grad_gemm(dq, W[:E])
            if ctx.needs_input_grad[1]:
                dxk = input_grad_gemm(dk, W[E:2 * E])
        dv = dv.contiguous()
        if ctx.needs_input_grad[2]:
            dxv = input_grad_gemm(dv, W[2 * E:])
        dW = db = None
        if ctx.needs_input_grad[3]:
            dW = grad_buffer_like(W)
            db = torch.empty(3 * E, device=W.device, dtype=W.dtype)
            for i, (dy, x) in enumerate(((dq, xq), (dk, xk), (dv, xv))):
                _dw_into(dy, x, dW[i * E:(i + 1) * E], db[i * E:(i + 1) * E], defer=ctx.defer)
        return dxq, dxk, dxv, dW, db, None, None


# ---- cross-attention with the K / V projections of a memory level merged over the layers that attend to it --------------------
# transformer_decoder.py:99-118 runs, for decoder layers l, l + 3, l + 6, the K and V projection of the SAME memory (level l)
# with three weights each: 18 skinny GEMMs per forward pass (M = 40 x 49 .. 784 tokens, N = 256), 18 input-gradient GEMMs whose
# results autograd adds up (12 accumulation kernels over up to 32 MB each) and 6 extra row-major copies of the memory.
# memory_kv computes K for the three layers with ONE GEMM (N = 3 x 256: 3 x the tiles on the small levels) and V with another,
# the attention backward of a layer writes dk / dv into its column block of one [tokens, 3 x 256] buffer (csrc/attention.hip
# strided outputs), and ONE input-gradient GEMM per operand (K = 768) contracts the whole buffer: nothing is concatenated and
# nothing accumulated.  The packed [3E, E] weight gradient of a layer is then filled by TWO nodes (rows 0:E by the layer's q
# projection, rows E:3E by the level's node) - _packed_grad.
_packed_grads = {}  # packed in_proj weight address -> (dW [3E, E], db [3E], who created it) inside a deferred_dw() context


def _packed_grad(W, who):
    """-> (dW, db, first): gradient tensors of a packed projection that two nodes fill.  Inside deferred_dw() (gradients are
    read after the context closes) the first caller creates them - dW is the registered flat-buffer view when there is one - and
    hands them to autograd, the second fills its rows and returns no gradient: autograd never adds anything.  Outside, every
    node returns a zero-padded tensor of its own and autograd adds the two (safe with any reader)."""
    if _dw_queue is None:
        return torch.zeros_like(W), torch.zeros(W.shape[0], device=W.device, dtype=W.dtype), True
    ent = _packed_grads.pop(W.data_ptr(), None)
    if ent is not None:
        return ent[0], ent[1], False
    dW, db = grad_buffer_like(W), torch.empty(W.shape[0], device=W.device, dtype=W.dtype)
    _packed_grads[W.data_ptr()] = (dW, db, who)
    return dW, db, True


def _finish_packed_grads():
    """deferred_dw() closes: a packed gradient only ONE of its two nodes reached (the other's output fed no loss) gets zeros
    in the rows nobody wrote"""
    for dW, db, who in _packed_grads.values():
        E = dW.shape[1]
        rows = slice(E, None) if who == "q" else slice(0, E)
        dW[rows].zero_()
        db[rows].zero_()
    _packed_grads.clear()


class _KVGrads:
    """dK / dV of one memory level: [rows, layers * E] each, allocated when the first attention backward asks for its block"""

    def __init__(self, rows, blocks, E, k_ptr, v_ptr, device):
        self.rows, self.n, self.E, self.k_ptr, self.v_ptr, self.device = rows, blocks, E, k_ptr, v_ptr, device
        self.dK = self.dV = None

    def matches(self, k, v, j):
        """k / v are exactly block j of the merged projections (not copies, not other tensors at a recycled address)"""
        return (k.data_ptr() == self.k_ptr + 4 * j * self.E and v.data_ptr() == self.v_ptr + 4 * j * self.E
                and tuple(k.shape) == (self.rows, self.E) and tuple(v.shape) == (self.rows, self.E)
                and k.stride(0) == self.n * self.E and v.stride(0) == self.n * self.E)

    def buffers(self):
        if self.dK is None:
            self.dK = torch.empty(self.rows, self.n * self.E, device=self.device, dtype=torch.float32)
            self.dV = torch.empty(self.rows, self.n * self.E, device=self.device, dtype=torch.float32)
        return self.dK, self.dV

    def blocks(self, j):
        dK, dV = self.buffers()
        E = self.E
        return dK[:, j * E:(j + 1) * E], dV[:, j * E:(j + 1) * E]


class _MemoryKV(Function):
    @staticmethod
    def forward(ctx, mem_k, mem_v, defer, Wk, bk, Wv, bv, *Wb):
        """mem_k / mem_v [rows, E]: key input (memory + position) and value input (memory) of one level; Wk / Wv [n E, E], bk /
        bv [n E]: the k / v rows of the n layers' packed projections, concatenated (copies, not differentiated); Wb = W_0, b_0,
        W_1, b_1, ...: the packed [3E, E] / [3E] parameters themselves -> k_0, v_0, k_1, v_1, ...: [rows, E] column blocks."""
        from . import attention as A
        defer, ctx.key_from_value = defer if isinstance(defer, tuple) else (defer, False)
        Ws = Wb[0::2]
        n, E = len(Ws), Ws[0].shape[1]
        K_all = forward_gemm(mem_k, Wk, bk, False)
        V_all = forward_gemm(mem_v, Wv, bv, False)
        if ctx.needs_input_grad[0]:
            expect_input_grad(*(W[E:2 * E] for W in Ws))  # the dX image is filled from the layers' slices: no concatenated weight
        if ctx.needs_input_grad[1]:
            expect_input_grad(*(W[2 * E:] for W in Ws))
        ctx.save_for_backward(mem_k, mem_v, *Ws)
        ctx.defer, ctx.n = defer, n
        ctx.holder = holder = _KVGrads(mem_k.shape[0], n, E, K_all.data_ptr(), V_all.data_ptr(), mem_k.device)
        outs = []
        slots = A.kv_gradient_slots
        for key in [key for key, (ref, _) in slots.items() if ref() is None]:
            del slots[key]  # holders of forward passes that never ran backward (their autograd graph is gone)
        for j in range(n):
            k_j, v_j = K_all[:, j * E:(j + 1) * E], V_all[:, j * E:(j + 1) * E]
            if any(ctx.needs_input_grad):
                slots[k_j.data_ptr()] = (weakref.ref(holder), j)
            outs += [k_j, v_j]
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        from . import attention as A
        mem_k, mem_v, *Ws = ctx.saved_tensors
        n, holder = ctx.n, ctx.holder
        E = Ws[0].shape[1]
        dK, dV = holder.buffers()
        for j in range(n):
            A.kv_gradient_slots.pop(holder.k_ptr + 4 * j * E, None)
            for g, buf in ((grads[2 * j], dK), (grads[2 * j + 1], dV)):
                dst = buf[:, j * E:(j + 1) * E]
                if g is None:
                    dst.zero_()  # (a layer whose attention output reached no loss)
                elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                    dst.copy_(g)  # a gradient that was not written in place by the strided attention backward
        dxk = input_grad_gemm(dK, tuple(W[E:2 * E] for W in Ws)) if ctx.needs_input_grad[0] else None
        if ctx.key_from_value and dxk is not None and ctx.needs_input_grad[1]:
            # mem_k = mem_v + (a constant): both gradients belong to mem_v - the second GEMM's epilogue sums them, the key
            # input's own edge carries nothing (autograd would add two [rows, E] tensors per level)
            dxv, dxk = input_grad_gemm(dV, tuple(W[2 * E:] for W in Ws), add=dxk if dxk.is_contiguous() else dxk.contiguous()), None
        else:
            dxv = input_grad_gemm(dV, tuple(W[2 * E:] for W in Ws)) if ctx.needs_input_grad[1] else None
        out = [dxk, dxv, None, None, None, None, None]
        for j, W in enumerate(Ws):
            if not ctx.needs_input_grad[7 + 2 * j]:
                out += [None, None]
                continue
            dW, db, first = _packed_grad(W, "kv")
            _dw_into(dK[:, j * E:(j + 1) * E], mem_k, dW[E:2 * E], db[E:2 * E], defer=ctx.defer)
            _dw_into(dV[:, j * E:(j + 1) * E], mem_v, dW[2 * E:], db[2 * E:], defer=ctx.defer)
            out += [dW, db] if first else [None, None]
        return tuple(out)


def memory_kv(mem_k, mem_v, level_params, defer=False, key_from_value=False):
    """The k / v projections of every cross-attention layer, one pair of GEMMs per memory level.
    key_from_value: the caller guarantees mem_k[l] = mem_v[l] + (a tensor that needs no gradient): the gradient of both inputs is
    then returned once, through mem_v.
    mem_k / mem_v: per level, the key input (memory + position) and the value input (memory), [B, hw, E];
    level_params: per level, [(in_proj_weight [3E, E], in_proj_bias [3E])] of the layers that attend to that level, in layer order
    -> per level, [(k, v)] as ROW VIEWS [B * hw, E] (row pitch layers * E) for ops.attention.attention."""
    E = mem_k[0].shape[-1]
    with torch.no_grad():  # k rows of level 0's layers, v rows of level 0's layers, k rows of level 1's, ...: ONE copy kernel each
        Wcat = torch.cat([W[r * E:(r + 1) * E] for params in level_params for r in (1, 2) for W, _ in params], 0)
        bcat = torch.cat([b[r * E:(r + 1) * E] for params in level_params for r in (1, 2) for _, b in params], 0)
    if FORWARD_PRECISION != "fp32" and _fwd_images is not None:
        _fwd_images.setdefault("pinned", []).append(Wcat)  # (bf16 images are cached by weight address for the step)
    out, off = [], 0
    for xk, xv, params in zip(mem_k, mem_v, level_params):
        n = len(params)
        if n == 0:  # fewer decoder layers than memory levels
            out.append([])
            continue
        flat = [t for pair in params for t in pair]
        rk, rv = slice(off, off + n * E), slice(off + n * E, off + 2 * n * E)
        off += 2 * n * E
        o = _MemoryKV.apply(xk.reshape(-1, E), xv.reshape(-1, E), (defer, key_from_value), Wcat[rk], bcat[rk], Wcat[rv], bcat[rv], *flat)
        out.append([(o[2 * j], o[2 * j + 1]) for j in range(n)])
    return out


class _InProjQ(Function):
    """q = x W[:E]^T + b[:E] of a packed nn.MultiheadAttention projection whose k / v rows are computed by memory_kv"""

    @staticmethod
    def forward(ctx, x2d, W, b, defer=False):
        E = W.shape[1]
        ctx.defer = defer
        q = forward_gemm(x2d, W[:E], b[:E], False)
        if ctx.needs_input_grad[0]:
            expect_input_grad(W[:E])
        ctx.save_for_backward(x2d, W)
        return q

    @staticmethod
    @once_differentiable
    def backward(ctx, dq):
        x2d, W = ctx.saved_tensors
        E = W.shape[1]
        dq = dq.contiguous()
        dx = input_grad_gemm(dq, W[:E]) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1]:
            dW, db, first = _packed_grad(W, "q")
            _dw_into(dq, x2d, dW[:E], db[:E], defer=ctx.defer)
            if not first:
                dW = db = None
        return dx, dW, db, None


def in_proj_q(xq, weight, bias, defer=False):
    """the q third of the packed projection ([..., E] -> [..., E]); k and v come from memory_kv"""
    E = weight.shape[1]
    return _InProjQ.apply(xq.reshape(-1, E), weight, bias, defer).view(*xq.shape[:-1], E)


def in_proj(xq, xk, xv, weight, bias, same_qk=False, defer=False):
    """Packed q/k/v projection of nn.MultiheadAttention ([..., E] inputs -> three [..., E] outputs; q and k of a
    self-attention layer are column blocks of one [rows, 2E] buffer).  same_qk: xq and xk are the same tensor."""
    E = weight.shape[1]
    if torch.is_autocast_enabled() or not xq.is_cuda or xq.dtype != torch.float32:
        return (linear(xq, weight[:E], bias[:E]), linear(xk, weight[E:2 * E], bias[E:2 * E]),
                linear(xv, weight[2 * E:], bias[2 * E:]))
    shp = (xq.shape[:-1], xk.shape[:-1], xv.shape[:-1])
    q, k, v = _InProj.apply(xq.reshape(-1, E), xk.reshape(-1, E), xv.reshape(-1, E), weight, bias, same_qk, defer)
    return q.view(*shp[0], E), k.view(*shp[1], E), v.view(*shp[2], E)


class _LinearCat(Function):
    """y = x @ cat(W1, W2)^T + cat(b1, b2): two nn.Linear layers that read the same input as ONE output buffer (two forward
    GEMMs into its column blocks - no concatenated weight copy; dX and dW once each over the merged gradient)."""

    @staticmethod
    def forward(ctx, x2d, w1, b1, w2, b2, defer=False):
        ctx.defer = defer
        n1, n2 = w1.shape[0], w2.shape[0]
        y = torch.empty(x2d.shape[0], n1 + n2, device=x2d.device, dtype=torch.float32)
        forward_gemm(x2d, w1, b1, False, out=y[:, :n1])
        forward_gemm(x2d, w2, b2, False, out=y[:, n1:])
        if ctx.needs_input_grad[0]:
            expect_input_grad(w1, w2)
        ctx.save_for_backward(x2d, w1, w2)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2d, w1, w2 = ctx.saved_tensors
        n1 = w1.shape[0]
        dy = dy.contiguous()
        dx = input_grad_gemm(dy, (w1, w2)) if ctx.needs_input_grad[0] else None
        dW = torch.empty(n1 + w2.shape[0], w1.shape[1], device=w1.device, dtype=w1.dtype)
        db = torch.empty(dW.shape[0], device=w1.device, dtype=w1.dtype)
        _dw_into(dy, x2d, dW, db, defer=ctx.defer)
        return dx, dW[:n1], db[:n1], dW[n1:], db[n1:], None


def linear_cat(x, w1, b1, w2, b2, defer=False):
    """[..., K] -> [..., N1 + N2]; fp32 CUDA tensors."""
    K = x.shape[-1]
    y = _LinearCat.apply(x.reshape(-1, K), w1, b1, w2, b2, defer)
    return y.view(*x.shape[:-1], y.shape[-1])


def _own_path(x, weight):
    return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and not torch.is_autocast_enabled()


def linear(x, weight, bias=None, relu=False, defer=False, mask_dx=False, grad_masked=False):
    """F.linear(x, weight, bias) [+ ReLU] on the head's kernels for fp32 CUDA tensors (any row count).
    mask_dx / grad_masked: see _Linear.forward (use `ffn`)."""
    K = x.shape[-1]
    N = weight.shape[0]
    if _own_path(x, weight):
        rows = x.numel() // K
        y = _Linear.apply(x.reshape(rows, K), weight, bias, relu, defer, mask_dx, grad_masked)
        return y.view(*x.shape[:-1], N)
    # autocast / non-fp32 callers (the host-PyTorch backbones' bf16 linears do not come through here: backbone_pvt._linear);
    # CPU tensors are the multi-process host-logic tests.  A CUDA caller landing here is told so, once.
    if x.is_cuda:
        _lib.fallback_notice("ops.linear.linear", f"x {x.dtype}, weight {weight.dtype}, autocast {torch.is_autocast_enabled()}: "
                             "the own GEMMs take fp32 operands outside autocast")
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.relu(y) if relu else y


def ffn(x, w1, b1, w2, b2, defer=True):
    """linear2(relu(linear1(x))) of the transformer FFN blocks (msdeformattn.py:125-134, transformer_decoder.py:178-182).
    The ReLU backward is folded into the second layer's input-gradient GEMM (its epilogue multiplies dH = dY . W2 by
    [H > 0]): no ReLU-gradient pass over the 1024- / 2048-wide hidden tensor (read dH + H, write dH)."""
    fused = _own_path(x, w1) and FFN_FUSED_RELU_GRAD and w2.shape[0] % 16 == 0
    h = linear(x, w1, b1, relu=True, defer=defer, grad_masked=fused)
    return linear(h, w2, b2, defer=defer, mask_dx=fused)


FFN_FUSED_RELU_GRAD = True  # the ReLU gradient of an FFN rides in linear2's input-gradient GEMM (False: a separate kernel)


class Linear(torch.nn.Linear):
    """nn.Linear whose forward/backward GEMMs run on the head's kernels (same parameters / state-dict names)."""

    defer_dw = False  # set by modules whose weight gradients may join the grouped launch (see deferred_dw)

    def forward(self, x):
        return linear(x, self.weight, self.bias, defer=self.defer_dw)
