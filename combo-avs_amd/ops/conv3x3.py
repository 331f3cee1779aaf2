"""3x3 / stride 1 / pad 1 convolution on channels_last fp32 maps as implicit GEMMs on the head's own MFMA kernels.

Replaces the cuDNN convolution behind the FPN output layer `layer_1` of the reference's pixel decoder
(pixel_decoder/msdeformattn.py:281-286, used at :349-352): 256 -> 256 channels on the 56 x 56 map, 3.7 GFLOP per frame,
the largest dense op of the head.  MIOpen's fp32 kernels needed 1.19 + 1.24 + 1.24 ms (forward, input
gradient, weight gradient) at 40 frames; here
  forward  Y  = conv(X, W)        csrc/gemm_f32.hip, CONV = true: exact fp32 MFMA (A rows gathered per tap, zero row for
                                  the padding) - forward values feed the decoder's mask thresholds
  dX          = conv(dY, W')      csrc/gemm_nt3.hip, CONV = true, W' = taps flipped, channels swapped (3-product bf16 split)
  dW          = dY^T . im2col(X)  csrc/gemm_tn.hip, CONV = true (3-product split, split-K over the tokens + the fused reduce)
No im2col buffer exists anywhere.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import os

from .. import _lib

ENABLED = True  # (module constant: tools flip it in-process for A/B measurements against MIOpen)
WGRAD_MAX_PARTIAL_MB = 64  # cap on the split-K partials of one weight-gradient launch in MiB (0 = the planner's split count); same-box A/B: profiles/r06_ab_conv_wgrad_partial_cap.txt


def usable(conv, x):
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and x.is_cuda and x.dtype == torch.float32
            and not torch.is_autocast_enabled() and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and x.shape[1] % 128 == 0 and conv.out_channels % 128 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2
            and x.shape[0] * x.shape[2] * x.shape[3] * max(x.shape[2], x.shape[3]) < 2 ** 31)


def _conv_tokens(x_tok, wm, bias, B, H, W, cin, cout, exact, relu=False):
    """x_tok [B*H*W, cin] (row stride free), wm [cout, 9*cin] -> [B*H*W, cout].  exact: fp32 MFMA (the forward value);
    else the 3-product bf16 split (the input gradient)."""
    y = torch.empty(B * H * W, cout, device=x_tok.device, dtype=torch.float32)
    assert B * H * W * cout * 4 < 2 ** 31 - 1  # (the kernels address Y with 32-bit byte offsets)
    lib, st = _lib.lib(), _lib.current_stream()
    from . import linear as L
    if exact and L.FORWARD_PRECISION != "fp32":  # the head's bf16 throughput mode: one bf16 product per multiply-add
        img = L.forward_image(wm)
        prev = lib.combo_gemm_nt2_products(L.forward_products())
        try:
            with _lib.timed("conv3x3_bf16", (B * H * W, cout, 9 * cin)):
                rc = lib.combo_conv3x3_nhwc_x3_pre_f32(x_tok.data_ptr(), x_tok.stride(0), img.data_ptr(), _lib.ptr(bias), y.data_ptr(),
                                                       cout, B, H, W, cin, cout, 1 if relu else 0, st)
        finally:
            lib.combo_gemm_nt2_products(prev)
        _lib.check(rc, "combo_conv3x3_nhwc_x3_pre_f32 (bf16 mode)")
        return y
    if exact:
        with _lib.timed("conv3x3_f32", (B * H * W, cout, 9 * cin)):
            rc = lib.combo_conv3x3_nhwc_f32(x_tok.data_ptr(), x_tok.stride(0), wm.data_ptr(), _lib.ptr(bias), y.data_ptr(), cout,
                                            B, H, W, cin, cout, 1 if relu else 0, st)
        _lib.check(rc, "combo_conv3x3_nhwc_f32")
        return y
    img = L.presplit(wm)
    with _lib.timed("conv3x3_x3", (B * H * W, cout, 9 * cin)):
        rc = lib.combo_conv3x3_nhwc_x3_pre_f32(x_tok.data_ptr(), x_tok.stride(0), img.data_ptr(), _lib.ptr(bias), y.data_ptr(),
                                               cout, B, H, W, cin, cout, 1 if relu else 0, st)
    _lib.check(rc, "combo_conv3x3_nhwc_x3_pre_f32")
    return y


def _wgrad_tokens(dy_tok, x_tok, B, H, W, cin, cout, ksize=3, stride=1):
    """-> dW in the parameter's own order [cout, cin, ksize, ksize] (contiguous): the split-K partials come out of the implicit GEMM as
    [cout, ky, kx, cin]; the finishing sum writes them permuted (csrc/gemm_tn.hip splitk_reduce_nchw_kernel).  H, W: the INPUT map
    (the rows of x_tok); dy_tok rows are the ceil(H / stride) x ceil(W / stride) output tokens."""
    lib = _lib.lib()
    taps = ksize * ksize
    M, K = B * (-(-H // stride)) * (-(-W // stride)), taps * cin
    splits = lib.combo_gemm_tn_splits(M, cout, K)
    if WGRAD_MAX_PARTIAL_MB:
        # (round 6) the planner aims at two workgroups per CU; for the few-token / wide layers (res5: 1 960 tokens, 512 x 4 608 outputs)
        # that is 8 partial copies of a 9.4 MB gradient - written and re-read cold inside the step (147 us for 9.25 GFLOP against 81 us
        # warm in tools/bench_conv_wgrad_splits.py).  Cap the bytes of partials instead.
        splits = max(1, min(splits, int(WGRAD_MAX_PARTIAL_MB * (1 << 20)) // (cout * K * 4)))
    mchunk = (-(-M // splits) + 15) // 16 * 16
    splits = -(-M // mchunk)
    part = torch.empty(splits, cout, K, device=x_tok.device, dtype=torch.float32)
    st = _lib.current_stream()
    with _lib.timed("conv3x3_wgrad_x3", (M, cout, K)):
        rc = lib.combo_conv_wgrad_x3_f32(dy_tok.data_ptr(), dy_tok.stride(0), x_tok.data_ptr(), x_tok.stride(0),
                                         part.data_ptr(), B, H, W, cin, cout, ksize, stride, splits, st)
    _lib.check(rc, "combo_conv_wgrad_x3_f32")
    dw = torch.empty(cout, cin, ksize, ksize, device=x_tok.device, dtype=torch.float32)
    _lib.check(lib.combo_splitk_reduce_nchw_f32(part.data_ptr(), splits, cout, taps, cin, dw.data_ptr(), st), "combo_splitk_reduce_nchw_f32")
    return dw


def _tokens(x):
    """NCHW view of channels_last memory -> [B*H*W, C] view (no copy)"""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C)


class _Conv3x3(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        wm = weight.permute(0, 2, 3, 1).reshape(cout, 9 * cin)  # [cout, ky, kx, cin]: one small copy per step
        y = _conv_tokens(_tokens(x), wm, bias, B, H, W, cin, cout, exact=True)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y.view(B, H, W, cout).permute(0, 3, 1, 2)  # NCHW view, channels_last memory

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dy_tok = _tokens(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX[t, ci] = sum_{tap, co} dY[t - shift(tap), co] W[co, ci, tap]: a convolution of dY with the flipped taps
            wt = weight.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout)
            dx = _conv_tokens(dy_tok, wt, None, B, H, W, cout, cin, exact=False).view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = _wgrad_tokens(dy_tok, _tokens(x), B, H, W, cin, cout)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy_tok.sum(0)
        return dx, dw, db


def conv3x3(x, weight, bias=None):
    return _Conv3x3.apply(x, weight, bias)
