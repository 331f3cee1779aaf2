"""The ResNet backbones' stride-1 convolutions on the head's 3-product GEMM kernels (fp32 recipe).

The reference's S4 / MS3 recipe trains two ResNet-50 encoders in fp32 (SOLVER.AMP.ENABLED False; detectron2's ResNet is not
part of /root/reference - backbones are SURVEY section 8 row f2).  The library's fp32 kernels ran them at ~95 TFLOP/s (the fp32
matrix instruction's rate) plus one bias / ReLU pass per convolution: 23 ms of a 55 ms step.  Here, for the 1x1 / stride 1 and
3x3 / stride 1 / pad 1 layers (46 of the 53 convolutions of a ResNet-50):
  * forward (FWD_X3): csrc/gemm_nt3.hip - the token GEMM / the implicit-GEMM 3x3 - with the 3-product bf16 split (every fp32
    product as hi.hi + lo.hi + hi.lo of bf16 pairs, fp32 accumulation: ~2^-17 relative per product, max error 5e-6 of the output
    range on these layers, tools/bench_r50_x3.py; the reference's own GPU path runs them through cuDNN with TF32 allowed, 2^-11),
    FrozenBN folded into the weights and bias (+ identity / shortcut branch) + ReLU in the GEMM epilogue: no separate pass;
  * input gradient (DX_OWN): the same kernels on dY with the transposed / tap-flipped weight image, the ReLU gradient of the
    producing layer in the epilogue where that layer has one consumer;
  * weight gradient: 1x1 as one PROBLEM of the grouped, deferred weight-gradient launch (ops.linear.weight_grad,
    csrc/gemm_tn.hip); 3x3 with >= 128 channels on the implicit-GEMM kernel of ops/conv3x3.py (no im2col buffer).
The bf16 hi/lo images of all weights of a backbone (forward and input-gradient form) are made by grouped launches per step
(weight_images).  Everything else (7x7 stem, stride-2 layers) keeps the library's kernels and autograd's backward."""
import ctypes

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from . import conv3x3 as C3
from . import linear as L

# Module constants (tests and tools/ flip them in-process; the measured A/B results are in DESIGN.md section 4):
ENABLED = True   # weight gradients of the stride-1 1x1 / 3x3 convolutions on the head's grouped kernels (-1.9 ms per step)
# input gradients on the head's 3-product kernels as well: bit 0 the 3x3 convolutions, bit 1 the 1x1 convolutions with >= DX_MIN_C
# channels
DX_OWN = 3
FWD_X3 = True    # forward on the 3-product kernels with the fused bias / residual / ReLU epilogue (needs weight_images)
DX_MIN_C = 64
# ReLU-gradient mask of conv1's output in the epilogue of the 3x3 convolution's own input-gradient kernel (needs DX_OWN bit 0)
MASK_3X3 = True
MASK_1X1 = True  # the same for conv2's output in conv3's (1x1) input-gradient GEMM
# (round 5) weight gradients that were the library's until now, on the generalised implicit TN GEMM (csrc/gemm_tn.hip
# combo_conv_wgrad_x3_f32): the 64-channel 3x3 layers of res2, and the stride-2 3x3 / 1x1 shortcut layers of res3.0 / res4.0 / res5.0
WGRAD_ANY_C = True
WGRAD_S2 = True
# input gradient of the 1x1 stride-2 shortcut convolutions: GEMM over the output tokens on the 3-product kernel + one expansion
# pass (csrc/biasact.hip expand_stride2) instead of the library's backward-data kernel
DX_S2_1X1 = True


# CU budget of the launches issued for the convolutions of ONE backbone while the Siam pair runs side by side on two streams
# (meta_arch.MaskFormer.parallel_backbones; csrc/abi.hip combo_set_cu_limit).  The forward pass records the budget in the autograd
# node, the backward pass (issued by autograd's thread on the forward's stream) applies the same one.
_cu_scope = 0


class backbone_cus:
    def __init__(self, n):
        self.n = int(n or 0)

    def __enter__(self):
        global _cu_scope
        self.prev, _cu_scope = _cu_scope, self.n

    def __exit__(self, *exc):
        global _cu_scope
        _cu_scope = self.prev


# (tools/ab_const.py) split counts of the backbone launches by token count, overriding the C planners: {("c", M): 1 | 3 | 9} for the 3x3
# tap split, {("g", M, cout, cin): s} for the K split of a 1x1 layer - the planners' cost models were fitted on warm micro-benchmarks
SPLIT_OVERRIDE = {}


def _plan3x3(lib, M, cout, cin):
    return SPLIT_OVERRIDE.get(("c", M)) or lib.combo_conv3x3_x3_splitk_plan(M, cout, cin)


def _plan1x1(lib, M, cout, cin):
    return SPLIT_OVERRIDE.get(("g", M, cout, cin)) or lib.combo_gemm_nt_x3_splitk_plan(M, cout, cin)


def kind(x, w, stride, padding):
    """0: not handled; 1: 1x1 stride 1; 3: 3x3 stride 1 pad 1; 21 / 23: the same with stride 2 (forward only)"""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and not torch.is_autocast_enabled()
            and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and torch.is_grad_enabled() and w.requires_grad):
        return 0
    return weight_kind(w, stride, padding, x.shape)


def weight_kind(w, stride, padding, x_shape=None):
    cout, cin, kh, kw = w.shape
    s = tuple(stride) if isinstance(stride, (tuple, list)) else (stride, stride)
    p = tuple(padding) if isinstance(padding, (tuple, list)) else (padding, padding)
    if (kh, kw) == (1, 1) and s == (1, 1) and p == (0, 0) and cin % 16 == 0 and cout % 16 == 0 and cin >= 64 and cout >= 64:
        return 1
    if (kh, kw) == (3, 3) and s == (1, 1) and p == (1, 1) and cin % 64 == 0 and cout % 64 == 0 and (
            x_shape is None or (x_shape[2] >= 2 and x_shape[3] >= 2 and x_shape[0] * x_shape[2] * x_shape[3] * max(x_shape[2], x_shape[3]) < 2 ** 31)):
        return 3
    if FWD_X3 and s == (2, 2) and cin % 64 == 0 and cout % 64 == 0 and (x_shape is None or x_shape[0] * x_shape[2] * x_shape[3] < 2 ** 29):
        # the first block of res3 / res4 / res5: forward on the own kernel (reproducible from run to run - the library's kernels
        # for these shapes accumulate with atomics), backward the library's
        if (kh, kw) == (1, 1) and p == (0, 0):
            return 21
        if (kh, kw) == (3, 3) and p == (1, 1):
            return 23
    return 0


_problem_cache = {}  # (source / image addresses ...) -> ctypes problem array of one backbone's weight images


def weight_images(weights, geometry):
    """bf16 hi/lo images (combo_presplit_bf16x2_*) of the folded weights of one backbone, all by grouped launches:
    weights [cout, cin, k, k] fp32, geometry [(stride, padding)] -> per weight None (layer not handled) or
    (forward image [cout, k*k*cin] with K ordered (ky, kx, cin), input-gradient image [cin, k*k*cout] with the taps flipped).
    A 3x3 weight in NCHW order is ONE problem per image whose threads read the 9 contiguous taps of 8 (cout, cin) pairs."""
    kinds = [weight_kind(w, s, p) if (w.is_cuda and w.dtype == torch.float32) else 0 for w, (s, p) in zip(weights, geometry)]
    two = lambda k: k < 20 or (k == 21 and DX_S2_1X1)  # layers whose input gradient runs on the own kernel: a second image
    total = sum((2 if two(k) else 1) * w.numel() for w, k in zip(weights, kinds) if k)
    if total == 0:
        return [None] * len(weights)
    buf = torch.empty(total, device=weights[0].device, dtype=torch.float32)
    out, off, spans = [], 0, []
    for w, k in zip(weights, kinds):
        if not k:
            out.append(None)
            continue
        cout, cin = w.shape[:2]
        n, ks = w.numel(), w.shape[2]
        if two(k):
            out.append((buf[off:off + n].view(cout, ks * ks * cin), buf[off + n:off + 2 * n].view(cin, ks * ks * cout)))
        else:
            out.append((buf[off:off + n].view(cout, ks * ks * cin), None))  # stride-2 3x3: forward only
        spans.append(w)
        off += (2 if two(k) else 1) * n
    # addresses recycle (the folded weights are fresh tensors every step): the key also carries everything else a problem encodes -
    # shape, strides (contiguous vs channels_last taps), the layer kind (decides which images exist, i.e. every offset) and FWD_X3
    key = (buf.data_ptr(), FWD_X3, DX_S2_1X1) + tuple((w.data_ptr(), tuple(w.shape), tuple(w.stride()), k) for w, k in zip(weights, kinds) if k)
    pr = _problem_cache.get(key)
    if pr is None:
        plist = []
        for w, kk, imgs in zip(weights, kinds, out):
            if not kk:
                continue
            cout, cin, k = w.shape[:3]
            s0, s1, s2, s3 = w.stride()
            fwd, dx = imgs
            if (s2, s3) == (k, 1) or k == 1:  # the taps of a (cout, cin) pair are contiguous: one problem per image
                plist.append(L._SplitProblem(w.data_ptr(), fwd.data_ptr(), s0, s1, k * k * cin, cout, cin, k * k, 0))
                if dx is not None:  # dX[t] = sum_tap' dY[t + shift(tap')] . W[.., tap flipped]
                    plist.append(L._SplitProblem(w.data_ptr(), dx.data_ptr(), s1, s0, k * k * cout, cin, cout, k * k, 1))
                continue
            for ky in range(k):  # any other layout (channels_last weights): one strided problem per tap
                for kx in range(k):
                    tap = ky * k + kx
                    src = w.data_ptr() + 4 * (ky * s2 + kx * s3)
                    plist.append(L._SplitProblem(src, fwd.data_ptr() + 4 * tap * cin, s0, s1, k * k * cin, cout, cin, 0, 0))
                    if dx is not None:
                        flip = (k - 1 - ky) * k + (k - 1 - kx)
                        plist.append(L._SplitProblem(src, dx.data_ptr() + 4 * flip * cout, s1, s0, k * k * cout, cin, cout, 0, 0))
        pr = (L._SplitProblem * len(plist))(*plist)
        if len(_problem_cache) > 64:
            _problem_cache.clear()
        _problem_cache[key] = pr
    _lib.check(_lib.lib().combo_presplit_bf16x2_grouped_f32(ctypes.cast(pr, ctypes.c_void_p), len(pr), _lib.current_stream()),
               "combo_presplit_bf16x2_grouped_f32")
    return out


def _x3_forward(x, k, img, bias, residual, relu):
    """relu(conv(x, W) + bias (+ residual)) on csrc/gemm_nt3.hip; x NCHW view of channels_last memory -> the same form"""
    B, cin, H, W = x.shape
    cout = img.shape[0]
    if k > 20:  # stride 2
        ho, wo = (H + 1) // 2, (W + 1) // 2
        y = torch.empty((B, cout, ho, wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        lib, M, ks = _lib.lib(), B * ho * wo, k - 20
        splits = _plan3x3(lib, M, cout, cin) if ks == 3 else 1
        ws = torch.empty(splits, M, cout, device=x.device, dtype=torch.float32) if splits > 1 else None
        x_tok, y_tok = C3._tokens(x), C3._tokens(y)
        aux = C3._tokens(residual) if residual is not None else None
        with _lib.timed("conv3x3_x3" if ks == 3 else "gemm_nt_x3", (M, cout, ks * ks * cin)):
            rc = lib.combo_conv_nhwc_x3_epi_f32(x_tok.data_ptr(), x_tok.stride(0), img.data_ptr(), _lib.ptr(bias), _lib.ptr(aux),
                                                1 if aux is not None else 0, y_tok.data_ptr(), y_tok.stride(0), B, H, W, cin, cout, ks, 2,
                                                1 if relu else 0, splits, _lib.ptr(ws), _lib.current_stream())
        _lib.check(rc, "combo_conv_nhwc_x3_epi_f32")
        return y
    y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    _x3_tokens(C3._tokens(x), k, img, bias, C3._tokens(residual) if residual is not None else None, 1, relu, C3._tokens(y), B, H, W, cin, cout)
    return y


def _x3_tokens(x_tok, k, img, bias, aux, aux_mode, relu, y_tok, B, H, W, cin, cout):
    lib, st = _lib.lib(), _lib.current_stream()
    M = B * H * W
    if k == 1:
        splits = _plan1x1(lib, M, cout, cin)
        ws = torch.empty(splits, M, cout, device=x_tok.device, dtype=torch.float32) if splits > 1 else None
        with _lib.timed("gemm_nt_x3", (M, cout, cin)):
            rc = lib.combo_gemm_nt_x3_epi_f32(x_tok.data_ptr(), x_tok.stride(0), img.data_ptr(), _lib.ptr(bias), _lib.ptr(aux),
                                              aux_mode if aux is not None else 0, y_tok.data_ptr(), y_tok.stride(0), M, cout, cin,
                                              1 if relu else 0, splits, _lib.ptr(ws), st)
        _lib.check(rc, "combo_gemm_nt_x3_epi_f32")
    else:
        splits = _plan3x3(lib, M, cout, cin)
        ws = torch.empty(splits, M, cout, device=x_tok.device, dtype=torch.float32) if splits > 1 else None
        with _lib.timed("conv3x3_x3", (M, cout, 9 * cin)):
            rc = lib.combo_conv3x3_nhwc_x3_epi_f32(x_tok.data_ptr(), x_tok.stride(0), img.data_ptr(), _lib.ptr(bias), _lib.ptr(aux),
                                                   aux_mode if aux is not None else 0, y_tok.data_ptr(), y_tok.stride(0), B, H, W, cin,
                                                   cout, 1 if relu else 0, splits, _lib.ptr(ws), st)
        _lib.check(rc, "combo_conv3x3_nhwc_x3_epi_f32")
    return y_tok


class _ConvWrw(Function):
    @staticmethod
    def forward(ctx, x, w, k, mask_dx=False, images=None, bias=None, residual=None, relu=False, passthrough=False):
        """mask_dx: x is the ReLU output of the producing layer and feeds nothing else - the input gradient is returned already
        multiplied by [x > 0] (folded into the dX GEMM's epilogue); the producer then skips its ReLU-gradient pass
        (ops.biasact.bias_act(grad_masked=True)).  Set in pairs by backbone.Bottleneck.
        images (weight_images) + FWD_X3: the forward runs on the 3-product kernel and applies bias (+ residual) (+ ReLU) itself -
        the caller routes the gradient of that epilogue (ops.biasact.bias_act(precomputed=True)); residual: values only."""
        ctx.k, ctx.mask_dx = k, mask_dx
        ctx.img_dx = images[1] if images is not None else None
        ctx.save_for_backward(x, w)
        ctx.passthrough = passthrough
        ctx.cus = _cu_scope
        if images is not None and FWD_X3:
            with _lib.cu_limit(ctx.cus):
                y = _x3_forward(x, k, images[0], bias, residual, relu)
        else:
            assert bias is None and residual is None and not relu
            y = F.conv2d(x, w, None, 2 if k > 20 else 1, 1 if k % 10 == 3 else 0)
        # passthrough (with mask_dx, stride 1): x has a SECOND consumer (the block's identity branch) - it reads this node's second
        # output, an alias of x, so that its gradient arrives HERE and is summed in the input-gradient GEMM's epilogue, before the
        # ReLU mask: dx = (dy . W + d_alias) * [x > 0] - one GEMM instead of GEMM + (add, mask) pass (csrc/biasact.hip relu_grad2)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, d_alias=None):
        with _lib.cu_limit(ctx.cus):
            return _ConvWrw._backward(ctx, dy, d_alias)

    @staticmethod
    def _backward(ctx, dy, d_alias=None):
        x, w = ctx.saved_tensors
        if d_alias is not None and not d_alias.is_contiguous(memory_format=torch.channels_last):
            d_alias = d_alias.contiguous(memory_format=torch.channels_last)
        B, cin, H, W = x.shape
        cout = w.shape[0]
        pad = 1 if ctx.k % 10 == 3 else 0
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        if ctx.k > 20:  # stride 2 (the first block of res3 / res4 / res5): input gradient the library's, weight gradient own
            assert not ctx.mask_dx and not ctx.passthrough
            # (the geometry limits of combo_conv_wgrad_x3_f32: maps of at least 2 x 2 input pixels, int32 token indices - anything
            #  else goes to the library instead of raising COMBO_EINVAL out of a backward pass)
            own_dw = (WGRAD_S2 and ctx.needs_input_grad[1] and cin % 4 == 0 and cout >= 64 and cout % 4 == 0 and H >= 2 and W >= 2
                      and B * H * W < 2 ** 29)
            own_dx = DX_S2_1X1 and ctx.k == 21 and ctx.needs_input_grad[0] and ctx.img_dx is not None and cin % 4 == 0
            dx, dw = None, None
            if (ctx.needs_input_grad[0] and not own_dx) or (ctx.needs_input_grad[1] and not own_dw):
                dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, (2, 2), (pad, pad), (1, 1), False, (0, 0), 1,
                                                                (ctx.needs_input_grad[0] and not own_dx, ctx.needs_input_grad[1] and not own_dw, False))
            if own_dx:
                dxc = L.gemm_nt_x3(C3._tokens(dy), ctx.img_dx, img=ctx.img_dx)  # [B * Ho * Wo, cin]: the gradient at the even pixels
                dx = torch.empty((B, cin, H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
                _lib.check(_lib.lib().combo_expand_stride2_f32(dxc.data_ptr(), dx.data_ptr(), B, H, W, cin, _lib.current_stream()),
                           "combo_expand_stride2_f32")
            if own_dw:
                dw = C3._wgrad_tokens(C3._tokens(dy), C3._tokens(x), B, H, W, cin, cout, ksize=3 if ctx.k == 23 else 1, stride=2)
            return dx, dw, None, None, None, None, None, None, None
        dx = dw = None
        if ctx.needs_input_grad[0] and ctx.k == 3 and (DX_OWN & 1) and ctx.img_dx is not None and d_alias is None:
            dx = torch.empty((B, cin, H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
            _x3_tokens(C3._tokens(dy), 3, ctx.img_dx, None, C3._tokens(x) if ctx.mask_dx else None, 2, False, C3._tokens(dx), B, H, W, cout, cin)
        elif ctx.needs_input_grad[0] and ctx.k == 1 and (DX_OWN & 2) and min(cin, cout) >= DX_MIN_C:
            mask = C3._tokens(x) if ctx.mask_dx else None
            add = C3._tokens(d_alias) if d_alias is not None else None
            if ctx.img_dx is not None:
                dx = L.gemm_nt_x3(C3._tokens(dy), ctx.img_dx, relu_mask=mask, img=ctx.img_dx, add=add)
            else:
                dx = L.input_grad_gemm(C3._tokens(dy), w.view(cout, cin), relu_mask=mask, add=add)
            dx = dx.view(B, H, W, cin).permute(0, 3, 1, 2)
        elif ctx.needs_input_grad[0]:  # input gradient: the library's kernel
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (pad, pad), (1, 1), False, (0, 0), 1,
                                                     (True, False, False))[0]
            if d_alias is not None:
                dx = dx + d_alias
            if ctx.mask_dx:
                dx = L.relu_grad(C3._tokens(dx.contiguous(memory_format=torch.channels_last)), C3._tokens(x)).view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dy_tok, x_tok = C3._tokens(dy), C3._tokens(x)
            if ctx.k == 1:
                g, _ = L.weight_grad(w.view(cout, cin), dy_tok, x_tok, False, True)  # joins the grouped launch when it can
                dw = g.view(cout, cin, 1, 1)
            elif (cin % 128 == 0 and cout % 128 == 0) or (WGRAD_ANY_C and cin % 4 == 0 and cout >= 64 and cout % 4 == 0):
                dw = C3._wgrad_tokens(dy_tok, x_tok, B, H, W, cin, cout)  # (round 5: any channel count - res2's 64-channel layers)
            else:  # the library's weight gradient
                dw = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (pad, pad), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1]
        return dx, dw, None, None, None, None, None, None, None


class _MaskedInput(Function):
    """identity whose gradient is multiplied by [x > 0]: the explicit ReLU-gradient pass for a `mask_dx` request on a layer
    that no own input-gradient kernel takes (the producer was told grad_masked=True and skips its own pass)"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return dy * (x > 0).to(dy.dtype)


def _library_notice(site, x, w, stride):
    """one stderr line per (site, layer geometry): this convolution runs on MIOpen.  By design for the backbones' 7x7 stems
    (3 input channels) - DESIGN.md section 6b; anything else listed is a layout / dtype condition of kind() that was not met."""
    _lib.fallback_notice(f"ops.convwrw.{site}[w {tuple(w.shape)}, stride {stride}]",
                         f"x {x.dtype} {'channels_last' if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) else 'NCHW'}, "
                         f"grad {torch.is_grad_enabled() and w.requires_grad}, autocast {torch.is_autocast_enabled()}")


def _forward_ok(x, w, residual):
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and not torch.is_autocast_enabled() and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)
            and (residual is None or (residual.dtype == torch.float32 and residual.shape[1] == w.shape[0]
                                      and residual.is_contiguous(memory_format=torch.channels_last))))


def conv_bias_act(x, w, bias, stride, padding, images=None, residual=None, relu=True, fanout=False, grad_masked=False, mask_dx=False,
                  passthrough=False):
    """relu(conv2d(x, w) + bias (+ residual)) of a FrozenBN-folded backbone convolution (bias: fp32 [cout] or None = no epilogue
    at all).  With `images` (weight_images) the whole expression is ONE launch of the 3-product kernel; otherwise the library's
    convolution followed by the fused bias / ReLU pass (ops/biasact.py).  fanout / grad_masked / mask_dx: see
    ops.biasact.bias_act and _ConvWrw.forward.  passthrough: x is a ReLU output whose producer skips its ReLU-gradient pass, with
    TWO consumers - this convolution and the reader of the second return value (an alias of x); the two gradients are summed and
    masked in this layer's input-gradient GEMM where it is an own kernel, else by ops.biasact.masked_fan -> (y, x_alias)."""
    from .biasact import bias_act, fusable
    if passthrough:
        if not (torch.is_grad_enabled() and x.requires_grad):
            return conv_bias_act(x, w, bias, stride, padding, images, residual, relu, fanout, grad_masked, False), x
        k = kind(x, w, stride, padding)
        if (k == 1 and images is not None and FWD_X3 and bias is not None and residual is None and fanout is False
                and (DX_OWN & 2) and min(w.shape[0], w.shape[1]) >= DX_MIN_C):
            z, x_alias = _ConvWrw.apply(x, w, k, True, images, bias, None, relu, True)
            assert fusable(z, None)
            return bias_act(z, bias, None, relu, False, grad_masked, precomputed=True), x_alias
        from .biasact import masked_fan
        x1, x2 = masked_fan(x)
        return conv_bias_act(x1, w, bias, stride, padding, images, residual, relu, fanout, grad_masked, False), x2
    if images is not None and FWD_X3 and not (torch.is_grad_enabled() and (w.requires_grad or x.requires_grad)) and _forward_ok(x, w, residual):
        # no gradient (evaluation): the same kernel without an autograd node - the features of an evaluation forward are
        # bit-identical to the training forward's
        k = weight_kind(w, stride, padding, x.shape)
        if k:
            with _lib.cu_limit(_cu_scope):
                y = _x3_forward(x, k, images[0], bias, residual, relu and bias is not None)
            return (y,) * max(int(fanout), 2) if (fanout and bias is not None) else y
    k = kind(x, w, stride, padding)
    if k and images is not None and FWD_X3 and (residual is None or (residual.dtype == torch.float32 and residual.shape[1] == w.shape[0]
                                                                     and residual.is_contiguous(memory_format=torch.channels_last))):
        z = _ConvWrw.apply(x, w, k, mask_dx, images, bias, residual.detach() if residual is not None else None, relu and bias is not None)
        if bias is None:
            return z
        assert fusable(z, residual)
        return bias_act(z, bias, residual, relu, fanout, grad_masked, precomputed=True)
    if mask_dx and not k:  # the caller planned on an own dX kernel that does not take this layer after all (a layout / size
        # condition of kind()): unmasked library dX + an explicit ReLU-gradient pass instead of an assertion
        x, mask_dx = _MaskedInput.apply(x), False
    if not k and x.is_cuda:
        _library_notice("conv_bias_act", x, w, stride)
    y = _ConvWrw.apply(x, w, k, mask_dx, images) if k else F.conv2d(x, w, None, stride, padding)
    if bias is None:
        return y
    return bias_act(y, bias, residual, relu, fanout, grad_masked)


def conv2d(x, w, stride, padding, mask_dx=False):
    """F.conv2d(x, w, None, stride, padding) whose weight gradient (and, for the 1x1 layers, input gradient) runs on the head's
    kernels where they apply.  mask_dx (only with kind(...) != 0): see _ConvWrw.forward."""
    k = kind(x, w, stride, padding)
    if k:
        return _ConvWrw.apply(x, w, k, mask_dx)
    if mask_dx:
        x = _MaskedInput.apply(x)
    if x.is_cuda:
        _library_notice("conv2d", x, w, stride)
    return F.conv2d(x, w, None, stride, padding)
