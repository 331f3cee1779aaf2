"""dst_i = cast(src_i * scale_i[channel]) for many tensors in one launch (csrc/foldcast.hip)."""
import ctypes

import torch

from .. import _lib


class _FoldProblem(ctypes.Structure):  # combo_fold_problem (include/combo_avs.h)
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("numel", ctypes.c_longlong),
                ("inner", ctypes.c_int), ("src_bf16", ctypes.c_int), ("dst_bf16", ctypes.c_int)]


def _code(t):
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float32:
        return 0
    raise RuntimeError("fold_cast: float32 / bfloat16 tensors only")


def fold_cast(srcs, dsts, scales=None):
    """srcs / dsts: lists of contiguous CUDA tensors of equal shapes (fp32 or bf16 each); scales: list of fp32 tensors with
    one value per leading-dimension slice (or None for a plain cast)."""
    n = len(srcs)
    if n == 0:
        return
    arr = (_FoldProblem * n)()
    for i, (s, d) in enumerate(zip(srcs, dsts)):
        if not (s.is_cuda and d.is_cuda and s.is_contiguous() and d.is_contiguous() and s.shape == d.shape):
            raise RuntimeError("fold_cast: contiguous CUDA tensors of equal shape expected")
        sc = None if scales is None else scales[i]
        inner = s.numel() // s.shape[0] if sc is not None else s.numel()
        arr[i] = _FoldProblem(s.data_ptr(), d.data_ptr(), 0 if sc is None else sc.data_ptr(), s.numel(), max(inner, 1), _code(s), _code(d))
    _lib.check(_lib.lib().combo_fold_cast_grouped(ctypes.cast(arr, ctypes.c_void_p), n, _lib.current_stream()),
               "combo_fold_cast_grouped")
