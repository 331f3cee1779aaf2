"""ctypes access to oracle/libmsda_oracle.so (the plain-C core-op oracle).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmsda_oracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(value, shapes, loc, aw):
    """numpy arrays (float32 or float64) -> out [B,Lq,M*D]"""
    suf = "f32" if value.dtype == np.float32 else "f64"
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    sh = np.ascontiguousarray(np.asarray(shapes, dtype=np.int64))
    lsi = np.concatenate([[0], np.cumsum(sh[:, 0] * sh[:, 1])[:-1]]).astype(np.int64)
    value, loc, aw = (np.ascontiguousarray(a) for a in (value, loc, aw))
    out = np.empty((B, Lq, M * D), dtype=value.dtype)
    getattr(lib(), "msda_core_forward_" + suf)(_p(value), _p(sh), _p(lsi), _p(loc), _p(aw), B, S, M, D, L, Lq, P, _p(out))
    return out


def backward(gout, value, shapes, loc, aw):
    suf = "f32" if value.dtype == np.float32 else "f64"
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    sh = np.ascontiguousarray(np.asarray(shapes, dtype=np.int64))
    lsi = np.concatenate([[0], np.cumsum(sh[:, 0] * sh[:, 1])[:-1]]).astype(np.int64)
    gout, value, loc, aw = (np.ascontiguousarray(a) for a in (gout, value, loc, aw))
    gv, gl, gw = np.empty_like(value), np.empty_like(loc), np.empty_like(aw)
    getattr(lib(), "msda_core_backward_" + suf)(_p(gout), _p(value), _p(sh), _p(lsi), _p(loc), _p(aw), B, S, M, D, L, Lq, P,
                                                _p(gv), _p(gl), _p(gw))
    return gv, gl, gw
