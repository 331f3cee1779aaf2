"""Backward of the host-PyTorch backbone convolutions on the head's gradient GEMM kernels (fp32 recipe).

The reference's S4 / MS3 recipe trains the two ResNet-50 encoders in fp32 (SOLVER.AMP.ENABLED False); with fp32 activations
MIOpen's backward-data and backward-weight kernels run on the fp32 matrix instruction (1/16 of the bf16 MFMA rate) and were
14 ms of a 64 ms step.  Policy of this package (ops/linear.py): FORWARD values in exact fp32, GRADIENTS with the 3-product
bf16 split (2^-17 per product).  So the forward convolution stays the library's fp32 kernel and the two gradients of

  * 1x1 / stride 1 convolutions (a token-major GEMM on channels_last maps): dX = dY . W on csrc/gemm_nt2.hip, dW = dY^T . X on
    csrc/gemm_tn.hip (split-K over the tokens),
  * 3x3 / stride 1 / pad 1 convolutions with >= 128 channels: the implicit-GEMM kernels of ops/conv3x3.py,

run here; every other convolution (7x7 stem, stride-2 layers, 64-channel 3x3) keeps autograd's library backward.
detectron2's ResNet [d2] is not part of /root/reference (backbones are outside SURVEY section 8's path)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import conv3x3 as C3
from . import linear as L


def _tokens(x):
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C)


def kind(x, w, stride, padding):
    """0: not handled; 1: 1x1 stride 1; 3: 3x3 stride 1 pad 1 (channel counts the implicit-GEMM weight-gradient kernel takes)"""
    if not (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and not torch.is_autocast_enabled()
            and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and torch.is_grad_enabled()):
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


class _ConvGrad(Function):
    @staticmethod
    def forward(ctx, x, w, k):
        ctx.k = k
        ctx.save_for_backward(x, w)
        return F.conv2d(x, w, None, 1, 1 if k == 3 else 0)  # forward: the library's fp32 convolution

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = w.shape[0]
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dy_tok, x_tok = _tokens(dy), _tokens(x)
        dx = dw = None
        if ctx.k == 1:
            w2d = w.view(cout, cin)
            if ctx.needs_input_grad[0]:
                dx = L.input_grad_gemm(dy_tok, w2d).view(B, H, W, cin).permute(0, 3, 1, 2)
            if ctx.needs_input_grad[1]:
                # computed at once (not deferred): the FrozenBN fold's backward consumes it inside this backward pass
                g, _ = L._dw_now(dy_tok, x_tok, False)
                dw = g.reshape(cout, cin, 1, 1)
            return dx, dw, None
        if ctx.needs_input_grad[0]:
            wt = w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout)
            dx = C3._conv_tokens(dy_tok, wt, None, B, H, W, cout, cin, exact=False).view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = C3._wgrad_tokens(dy_tok, x_tok, B, H, W, cin, cout).permute(0, 3, 1, 2)
            if not dw.is_contiguous():
                dw = dw.contiguous()
        return dx, dw, None


def conv2d(x, w, stride, padding):
    """F.conv2d(x, w, None, stride, padding) whose gradients run on the head's 3-product bf16 kernels where they apply"""
    k = kind(x, w, stride, padding)
    if k:
        return _ConvGrad.apply(x, w, k)
    return F.conv2d(x, w, None, stride, padding)
