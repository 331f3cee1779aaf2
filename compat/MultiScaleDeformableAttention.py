"""`MultiScaleDeformableAttention` - the name the reference imports for its native op
(models/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:21 `import MultiScaleDeformableAttention as MSDA`;
pybind definitions at ops/src/vision.cpp:18-21, declarations at ops/src/ms_deform_attn.h:25-66).

Put `<repo>/compat` on PYTHONPATH in place of the reference's compiled extension and its unchanged `MSDeformAttnFunction`
(ms_deform_attn_func.py:32-50) calls

    MSDA.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    MSDA.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step)

positionally, exactly as the pybind module defines them.  This file is a plain ctypes binding of the C ABI
(include/combo_avs.h: combo_msda_{forward,backward}_{f32,f64}) - it imports nothing from combo_avs_amd, so it is what a
maintainer of the reference would add.  Library location: $COMBO_AVS_LIB, else
<repo>/combo-avs_amd/lib/libcombo_avs_hip.so next to this file's parent directory.  No CPU path: CPU tensors raise
(the reference's CPU stub raises AT_ERROR("Not implement on cpu"), ops/src/cpu/ms_deform_attn_cpu.cpp:20-38).
"""
import ctypes
import os

import torch  # load the HIP runtime torch uses BEFORE the kernel library (one runtime per process)

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("COMBO_AVS_LIB") or os.path.join(os.path.dirname(_HERE), "combo-avs_amd", "lib", "libcombo_avs_hip.so")
if not os.path.exists(_PATH):
    raise ImportError("MultiScaleDeformableAttention: %s not found - build it with `python combo-avs_amd/build.py` or set "
                      "COMBO_AVS_LIB (there is no CPU fallback)" % _PATH)
_lib = ctypes.CDLL(_PATH)
_vp, _i = ctypes.c_void_p, ctypes.c_int
for _sfx in ("f32", "f64"):
    _f = getattr(_lib, "combo_msda_forward_" + _sfx)
    _f.argtypes, _f.restype = [_vp] * 5 + [_i] * 7 + [_vp, _i, _vp], _i
    _b = getattr(_lib, "combo_msda_backward_" + _sfx)
    _b.argtypes, _b.restype = [_vp] * 6 + [_i] * 7 + [_vp] * 3 + [_i, _vp], _i
_SFX = {torch.float32: "f32", torch.float64: "f64"}


def _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    # the c10 asserts of ops/src/cuda/ms_deform_attn_cuda.cu:33-43: contiguous CUDA tensors
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor (Not implement on cpu)" % name)
    if value.dtype not in _SFX or sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("ms_deform_attn: float32 / float64 tensors of one dtype expected")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("ms_deform_attn: spatial_shapes / level_start_index must be int64")
    B, S, M, D = value.shape
    B2, Lq, M2, L, P, two = sampling_loc.shape
    if B2 != B or M2 != M or two != 2 or tuple(attn_weight.shape) != (B, Lq, M, L, P) or tuple(spatial_shapes.shape) != (L, 2) \
            or tuple(level_start_index.shape) != (L,):
        raise RuntimeError("ms_deform_attn: inconsistent shapes")
    return B, S, M, D, L, Lq, P


def _rc(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError_t / COMBO error %d" % (what, rc))


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """-> output [B, Lq, M*D] (ms_deform_attn_cuda_forward, ops/src/cuda/ms_deform_attn_cuda.cu:25-85).  im2col_step only chunks the
    batch in the reference; here the batch is never chunked and the value is ignored."""
    B, S, M, D, L, Lq, P = _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        _rc(getattr(_lib, "combo_msda_forward_" + _SFX[value.dtype])(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), B, S, M, D, L, Lq, P, out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream),
            "combo_msda_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (ms_deform_attn_cuda_backward, ms_deform_attn_cuda.cu:88-157; the
    gradient buffers are zero-filled here as at :126-128)."""
    B, S, M, D, L, Lq, P = _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if not grad_output.is_contiguous():
        raise RuntimeError("grad_output tensor has to be contiguous")
    if not grad_output.is_cuda or grad_output.dtype != value.dtype or grad_output.numel() != B * Lq * M * D:
        raise RuntimeError("ms_deform_attn_backward: grad_output [B, Lq, M*D] of value's dtype on the GPU expected")
    gv, gl, gw = torch.zeros_like(value), torch.zeros_like(sampling_loc), torch.zeros_like(attn_weight)
    with torch.cuda.device(value.device):
        _rc(getattr(_lib, "combo_msda_backward_" + _SFX[value.dtype])(
            grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, M, D, L, Lq, P, gv.data_ptr(), gl.data_ptr(), gw.data_ptr(), 0,
            torch.cuda.current_stream().cuda_stream), "combo_msda_backward")
    return [gv, gl, gw]
