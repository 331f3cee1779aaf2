"""Weight gradients of the ResNet backbones' stride-1 convolutions on the head's gradient GEMM kernels (fp32 recipe).

The reference's S4 / MS3 recipe trains two ResNet-50 encoders in fp32 (SOLVER.AMP.ENABLED False; detectron2's ResNet is not
part of /root/reference - backbones are SURVEY section 8 row f2).  The library's fp32 weight-gradient kernels (split-K with
atomics + a zero-fill launch per convolution) were 8.0 ms of a 59 ms step: 352 launches at ~80 TFLOP/s.  Policy of this
package (ops/linear.py): FORWARD values exact fp32, GRADIENTS with the 3-product bf16 split.  So forward and input gradient
stay the library's kernels and only dW moves:
  * 1x1 / stride 1: dW = dY^T . X over the channels_last token views - one PROBLEM of the grouped, deferred weight-gradient
    launch (ops.linear.weight_grad, csrc/gemm_tn.hip): all of them and the head's own run as one launch + one reduce;
  * 3x3 / stride 1 / pad 1 with >= 128 channels: the implicit-GEMM kernel of ops/conv3x3.py (no im2col buffer).
Everything else (7x7 stem, stride-2 layers, the 64-channel 3x3) keeps autograd's library backward."""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import conv3x3 as C3
from . import linear as L

# Module constants (tests and tools/ flip them in-process; the measured A/B results are in DESIGN.md section 4):
ENABLED = True   # weight gradients of the stride-1 1x1 / 3x3 convolutions on the head's grouped kernels (-1.9 ms per step)
# input gradients on the head's 3-product kernels as well: bit 0 the 3x3 convolutions (measured: +1.3 ms per step, off), bit 1 the
# 1x1 convolutions with >= DX_MIN_C channels (measured: -0.85 ms per step with all of them, on)
DX_OWN = 2
FWD_OWN = False  # forward of the 1x1 layers on the exact-fp32 GEMM without a fused epilogue: +0.4 ms per step, off
DX_MIN_C = 64


def kind(x, w, stride, padding):
    """0: not handled; 1: 1x1 stride 1; 3: 3x3 stride 1 pad 1"""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and not torch.is_autocast_enabled()
            and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and torch.is_grad_enabled() and w.requires_grad):
        return 0
    cout, cin, kh, kw = w.shape
    s = tuple(stride) if isinstance(stride, (tuple, list)) else (stride, stride)
    p = tuple(padding) if isinstance(padding, (tuple, list)) else (padding, padding)
    if (kh, kw) == (1, 1) and s == (1, 1) and p == (0, 0) and cin % 16 == 0 and cout % 16 == 0 and cin >= 64 and cout >= 64:
        return 1
    if (kh, kw) == (3, 3) and s == (1, 1) and p == (1, 1) and cin % 128 == 0 and cout % 128 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2 \
            and x.shape[0] * x.shape[2] * x.shape[3] * max(x.shape[2], x.shape[3]) < 2 ** 31:
        return 3
    return 0


class _ConvWrw(Function):
    @staticmethod
    def forward(ctx, x, w, k, mask_dx=False):
        """mask_dx: x is the ReLU output of the producing layer and feeds nothing else - the input gradient is returned already
        multiplied by [x > 0] (folded into the dX GEMM's epilogue); the producer then skips its ReLU-gradient pass
        (ops.biasact.bias_act(grad_masked=True)).  Set in pairs by backbone.Bottleneck."""
        ctx.k, ctx.mask_dx = k, mask_dx
        ctx.save_for_backward(x, w)
        if k == 1 and FWD_OWN and L.f32_ok(C3._tokens(x), w.view(w.shape[0], w.shape[1])):
            B, cin, H, W = x.shape
            y = torch.empty((B, w.shape[0], H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
            L.gemm_nt_f32(C3._tokens(x), w.view(w.shape[0], cin), out=C3._tokens(y))  # exact fp32 MFMA (csrc/gemm_f32.hip)
            return y
        return F.conv2d(x, w, None, 1, 1 if k == 3 else 0)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        pad = 1 if ctx.k == 3 else 0
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0] and ctx.k == 3 and (DX_OWN & 1):
            wt = w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout)
            dx = C3._conv_tokens(C3._tokens(dy), wt, None, B, H, W, cout, cin, exact=False).view(B, H, W, cin).permute(0, 3, 1, 2)
        elif ctx.needs_input_grad[0] and ctx.k == 1 and (DX_OWN & 2) and min(cin, cout) >= DX_MIN_C:
            dx = L.input_grad_gemm(C3._tokens(dy), w.view(cout, cin), relu_mask=C3._tokens(x) if ctx.mask_dx else None)
            dx = dx.view(B, H, W, cin).permute(0, 3, 1, 2)
        elif ctx.needs_input_grad[0]:  # input gradient: the library's kernel
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (pad, pad), (1, 1), False, (0, 0), 1,
                                                     (True, False, False))[0]
            if ctx.mask_dx:
                dx = L.relu_grad(C3._tokens(dx.contiguous(memory_format=torch.channels_last)), C3._tokens(x)).view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dy_tok, x_tok = C3._tokens(dy), C3._tokens(x)
            if ctx.k == 1:
                g, _ = L.weight_grad(w.view(cout, cin), dy_tok, x_tok, False, True)  # joins the grouped launch when it can
                dw = g.view(cout, cin, 1, 1)
            else:
                dw = C3._wgrad_tokens(dy_tok, x_tok, B, H, W, cin, cout).permute(0, 3, 1, 2)
                if not dw.is_contiguous():
                    dw = dw.contiguous()
        return dx, dw, None, None


def conv2d(x, w, stride, padding, mask_dx=False):
    """F.conv2d(x, w, None, stride, padding) whose weight gradient (and, for the 1x1 layers, input gradient) runs on the head's
    kernels where they apply.  mask_dx (only with kind(...) != 0): see _ConvWrw.forward."""
    k = kind(x, w, stride, padding)
    if k:
        return _ConvWrw.apply(x, w, k, mask_dx)
    assert not mask_dx
    return F.conv2d(x, w, None, stride, padding)
