"""ctypes binding of libcombo_avs_hip.so (the C ABI declared in include/combo_avs.h).

Loud by design: `lib()` raises RuntimeError when the shared library has not been built; nothing in
this package falls back to PyTorch/CPU for an op that has a HIP kernel.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcombo_avs_hip.so")
_lib = None

c_void_p, c_int, c_float, c_longlong = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_SIGNATURES = {
    "combo_abi_version": [],
    "combo_build_arch": [],
    "combo_set_cu_limit": [c_int],
    "combo_dwconv3x3_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_dwconv3x3_wgrad_slices": [c_int] * 4,
    "combo_dwconv3x3_wgrad_strips": [c_int],
    "combo_dwconv3x3_wgrad_finish_f32": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "combo_dwconv3x3_wgrad_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_fold_cast_grouped": [c_void_p, c_int, c_void_p],
    "combo_prenorm_forward": [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_float, c_longlong, c_int, c_void_p, c_void_p,
                              c_int, c_void_p, c_void_p, c_void_p],
    "combo_prenorm_backward": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p],
    "combo_bias_ln_bf16_forward": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "combo_bias_ln_bf16_backward": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p],
    "combo_colsum_grouped_slices": [c_longlong, c_int],
    "combo_colsum_grouped": [c_void_p, c_int, c_void_p],
    "combo_colsum_slices": [c_longlong, c_int, c_longlong],
    "combo_colsum": [c_void_p, c_longlong, c_int, c_longlong, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "combo_bias_act_bf16": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
    "combo_bias_act_f32": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
    "combo_relu_grad_bf16": [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p],
    "combo_relu_grad_f32": [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p],
    "combo_relu_grad2_f32": [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p],
    "combo_relu_grad3_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p],
    "combo_expand_stride2_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "combo_msda_backward_needs_zero": [c_int] * 6,
    "combo_msda_forward_f32": [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_int, c_void_p],
    "combo_msda_forward_f64": [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_int, c_void_p],
    "combo_msda_backward_f32": [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 3 + [c_int, c_void_p],
    "combo_msda_backward_f64": [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 3 + [c_int, c_void_p],
    "combo_msda_backward_win_ok": [c_void_p, c_int, c_int, c_int, c_int],
    "combo_msda_backward_win_f32": [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 3 + [c_void_p],
    "combo_event_create": [c_void_p],
    "combo_event_record": [c_void_p, c_void_p, c_int],
    "combo_event_elapsed_us": [c_void_p, c_void_p, c_void_p],
    "combo_event_destroy": [c_void_p],
    "combo_upsample2x_bilinear_nhwc_f32": [c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_upsample2x_bilinear_add_nhwc_f32": [c_void_p, c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_upsample2x_bilinear_nhwc_backward_f32": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_longlong, c_void_p],
    "combo_groupnorm_nhwc_slices": [c_int],
    "combo_groupnorm_nhwc_forward_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "combo_groupnorm_nhwc_backward_f32": [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 6,
    "combo_msda_prep_forward_f32": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "combo_msda_prep_backward_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_bifuse_chunks": [c_int, c_int],
    "combo_bifuse_forward_f32": [c_void_p] * 3 + [c_float] + [c_void_p] * 8 + [c_float, ctypes.c_ulonglong, c_void_p] + [c_int] * 4 + [c_void_p] * 7,
    "combo_bifuse_backward1_f32": [c_void_p] * 3 + [c_float] + [c_void_p] * 7 + [c_float, ctypes.c_ulonglong, c_void_p] + [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 5,
    "combo_bifuse_backward2_f32": [c_void_p] * 3 + [c_float] + [c_void_p] * 5 + [c_float, ctypes.c_ulonglong, c_void_p] + [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 5,
    "combo_matcher_cost_f32": [c_void_p] * 6 + [c_int] * 9 + [c_float] * 3 + [c_void_p] * 3,
    "combo_sem_mix": [c_int, c_int] + [c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 3,
    "combo_semantic_inference_f32": [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p],
    "combo_conv3x3_wgrad_x3_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p] + [c_int] * 6 + [c_void_p],
    "combo_conv_wgrad_x3_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p] + [c_int] * 8 + [c_void_p],
    "combo_presplit_bf16x2_f32": [c_void_p, c_longlong, c_longlong, c_int, c_int, c_void_p, c_void_p],
    "combo_presplit_bf16x2_grouped_f32": [c_void_p, c_int, c_void_p],
    "combo_presplit_bf16x2_batched_f32": [c_void_p, c_longlong, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_gemm_nt_x3_pre_batched_f32": [c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong, c_longlong,
                                         c_int, c_int, c_int, c_int, c_int, c_void_p],
    "combo_gemm_nt_x3_pre_masked_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p],
    "combo_downsample_tokens_f32": [c_void_p] + [c_int] * 6 + [c_void_p, c_void_p],
    "combo_mask_bits_f32": [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_int, c_void_p, c_void_p],
    "combo_mask_logits_all_f32": [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    "combo_timing_set_buffer": [c_void_p, c_int],
    "combo_timing_slots_used": [],
    "combo_timing_rewind": [],
    "combo_timing_truncated": [],
    "combo_timing_fold": [c_void_p],
    "combo_timing_slot_info": [c_int, c_void_p, c_void_p],
    "combo_timing_slot_bytes": [c_int, c_void_p],
    "combo_gemm_smallm_splits": [c_int, c_int, c_int],
    "combo_gemm_smallm_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_void_p] + [c_int] * 5 + [c_void_p],
    "combo_gemm_smallk_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong, c_longlong, c_int, c_int, c_void_p],
    "combo_gemm_tn_smalln_slices": [c_longlong],
    "combo_gemm_tn_smalln_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "combo_gemm_nt_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p],
    "combo_gemm_nt2_products": [c_int],
    "combo_presplit_pieces": [c_int],
    "combo_gemm_nt_x3_tile": [c_int],
    "combo_gemm_nt_x3_prof_buffer": [c_void_p],
    "combo_gemm_nt_x3_splitk_plan": [c_int, c_int, c_int],
    "combo_gemm_nt_x3_splitk_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_gemm_nt_splitk_plan": [c_int, c_int, c_int],
    "combo_gemm_nt_splitk_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int,
                                 c_int, c_void_p, c_void_p],
    "combo_gemm_nt_batched_f32": [c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong,
                                  c_int, c_int, c_int, c_int, c_int, c_void_p],
    "combo_conv3x3_nhwc_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong] + [c_int] * 6 + [c_void_p],
    "combo_wall_clock_khz": [],
    "combo_gemm_nt_x3_pre_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p],
    "combo_gemm_nt_x3_pre_splitk_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p],
    "combo_conv3x3_nhwc_x3_pre_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_longlong] + [c_int] * 6 + [c_void_p],
    "combo_gemm_nt_x3_epi_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_longlong] + [c_int] * 5 + [c_void_p, c_void_p],
    "combo_gemm_nt_x3_epi2_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong] + [c_int] * 5 + [c_void_p, c_void_p],
    "combo_conv3x3_x3_splitk_plan": [c_longlong, c_int, c_int],
    "combo_conv3x3_nhwc_x3_epi_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_longlong] + [c_int] * 7 + [c_void_p, c_void_p],
    "combo_conv_nhwc_x3_epi_f32": [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_longlong] + [c_int] * 9 + [c_void_p, c_void_p],
    "combo_gemm_tn_splits": [c_int, c_int, c_int],
    "combo_gemm_tn_x3_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "combo_gemm_tn_x3_grouped_f32": [c_void_p, c_int, c_void_p],
    "combo_splitk_reduce_grouped_f32": [c_void_p, c_int, c_void_p],
    "combo_ln_param_grad_grouped_f32": [c_void_p, c_int, c_void_p],
    "combo_add_layernorm_forward_f32": [c_void_p] * 4 + [c_float, c_longlong, c_int] + [c_void_p] * 5 + [c_longlong, c_void_p, c_void_p],
    "combo_layernorm_backward_f32": [c_void_p] * 5 + [c_longlong, c_int] + [c_void_p] * 6,
    "combo_splitk_reduce_f32": [c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "combo_splitk_reduce_nchw_f32": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "combo_uncertain_points_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "combo_mask_loss_forward_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "combo_mask_loss_backward_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p],
    "combo_cosine_stats_f32": [c_void_p, c_longlong, c_longlong, c_int, c_void_p, c_void_p, c_void_p],
    "combo_cosine_grad_f32": [c_void_p, c_longlong, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_void_p],
    "combo_lsap_small_f32": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "combo_attention_forward_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_void_p] + [c_int] * 5 + [c_float, c_void_p, c_void_p, c_void_p],
    "combo_attention_backward_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_void_p] + [c_int] * 5 + [c_float] + [c_void_p] * 8,
    "combo_attention_backward_ld_f32": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_void_p] + [c_int] * 5 + [c_float]
                                       + [c_void_p] * 6 + [c_longlong, c_void_p, c_longlong, c_void_p],
    "combo_adamw_f32": [c_void_p] * 4 + [c_longlong, c_void_p] + [c_float] * 7 + [c_void_p],
    "combo_sra_attention_ok": [c_int, c_int, c_int],
    "combo_sra_attention_backward_workspace": [c_int, c_int, c_int, c_int],
    "combo_sra_attention_forward_bf16": [c_void_p] * 4 + [c_int] * 4 + [c_float, c_void_p],
    "combo_sra_attention_backward_bf16": [c_void_p] * 9 + [c_int] * 4 + [c_float, c_void_p],
}
_RESTYPES = {"combo_build_arch": ctypes.c_char_p, "combo_sra_attention_backward_workspace": ctypes.c_longlong}


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; load it FIRST so that this library binds to the same
        # HIP runtime instance as torch's allocator/streams (same SONAME -> first one loaded wins).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP library not built: {LIB_PATH} is missing. Run `python combo-avs_amd/build.py` "
                "(or __graft_entry__.build()). There is no CPU fallback for the hot path.")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(_lib, name)  # AttributeError if the .so is stale
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
    return _lib


class HipError(RuntimeError):
    pass


def check(code, what):
    if code != 0:
        raise HipError(f"{what} failed with hipError_t {code}")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors, channels_last=False):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("combo_avs_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t is not None and not (t.is_contiguous(memory_format=torch_channels_last()) if channels_last and t.dim() == 4
                                  else t.is_contiguous()):
            raise RuntimeError("combo_avs_amd ops need contiguous tensors")


def torch_channels_last():
    import torch
    return torch.channels_last


# ---- kernel timing (bench.py): HIP events from the C ABI around selected launches -------------------------------
class cu_limit:
    """`with cu_limit(n):` the persistent GEMM launches issued by this thread inside the block use at most n CUs (csrc/abi.hip);
    n = 0 / None: no change"""

    def __init__(self, n):
        self.n = int(n or 0)

    def __enter__(self):
        self.prev = lib().combo_set_cu_limit(self.n) if self.n else None

    def __exit__(self, *exc):
        if self.n:
            lib().combo_set_cu_limit(self.prev)


_noticed = set()


def fallback_notice(site, why):
    """One line on stderr, once per call site and process, whenever an op of this package hands CUDA work to a LIBRARY kernel
    (ATen / hipBLASLt / MIOpen) because its operands do not meet the own kernel's conditions.  SURVEY 8(b): the reference's
    module-level `except` swallowed such routing silently (ms_deform_attn.py:123) and hid performance bugs; here the result is
    still correct, but the user is told which path ran.  COMBO_QUIET_FALLBACK=1 silences the notices."""
    if site in _noticed:
        return
    _noticed.add(site)
    if os.environ.get("COMBO_QUIET_FALLBACK") == "1":
        return
    import sys
    print(f"[combo_avs_amd] {site}: running on a library kernel, not the package's HIP kernel ({why})", file=sys.stderr, flush=True)


_timing = None  # {kind: [(start_event, stop_event, meta)]} while a measurement is running


def start_timing():
    """Bracket every instrumented launch (ops wrapped in `timed(kind)`) with HIP events on the launch stream until
    stop_timing().  Launches issued while the stream is being captured into a hipGraph are skipped (HIP rejects event
    records inside a capture on ROCm 7)."""
    global _timing
    _timing = {}


def stop_timing():
    """-> {kind: [(microseconds, meta), ...]} (synchronises)."""
    import torch
    global _timing
    t, _timing = _timing, None
    torch.cuda.synchronize()
    out = {}
    for kind, evs in (t or {}).items():
        out[kind] = []
        for s, e, meta in evs:
            us = ctypes.c_float(0.0)
            check(lib().combo_event_elapsed_us(s, e, ctypes.byref(us)), "combo_event_elapsed_us")
            out[kind].append((float(us.value), meta))
            lib().combo_event_destroy(s)
            lib().combo_event_destroy(e)
    return out


class timed:
    def __init__(self, kind, meta=None):
        self.kind, self.meta = kind, meta

    def __enter__(self):
        import torch
        self.on = _timing is not None and not torch.cuda.is_current_stream_capturing()
        if self.on:
            self.s, self.e = ctypes.c_void_p(), ctypes.c_void_p()
            check(lib().combo_event_create(ctypes.byref(self.s)), "combo_event_create")
            check(lib().combo_event_create(ctypes.byref(self.e)), "combo_event_create")
            check(lib().combo_event_record(self.s, current_stream(), 0), "combo_event_record")

    def __exit__(self, *a):
        if self.on and _timing is not None:
            check(lib().combo_event_record(self.e, current_stream(), 0), "combo_event_record")
            _timing.setdefault(self.kind, []).append((self.s, self.e, self.meta))
