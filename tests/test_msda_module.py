"""Boundary row b1: the shipped `MultiScaleDeformableAttention` module (compat/MultiScaleDeformableAttention.py) - the name the
reference's autograd wrapper imports (ops/functions/ms_deform_attn_func.py:21) with the two pybind functions of
ops/src/vision.cpp:18-21, called POSITIONALLY as ms_deform_attn_func.py:36-48 does."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _module():
    compat = os.path.join(ROOT, "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    return importlib.import_module("MultiScaleDeformableAttention")


def test_module_imports_by_the_reference_name_and_exports_both_functions():
    MSDA = _module()
    assert callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)
    # a plain ctypes binding of the C ABI: nothing of the product package is pulled in by the stub itself
    src = open(MSDA.__file__).read()
    assert "combo_avs_amd" not in src.split('"""', 2)[2] and "oracle" not in src.split('"""', 2)[2]
    import inspect
    assert list(inspect.signature(MSDA.ms_deform_attn_forward).parameters) == [
        "value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "im2col_step"]
    assert list(inspect.signature(MSDA.ms_deform_attn_backward).parameters) == [
        "value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "grad_output", "im2col_step"]


def test_cpu_tensors_raise_like_the_reference_cpu_stub():
    """ops/src/cpu/ms_deform_attn_cpu.cpp:20-38: AT_ERROR("Not implement on cpu") - no CPU fallback here either."""
    MSDA = _module()
    v = torch.zeros(1, 4, 2, 32)
    sh = torch.tensor([[2, 2]])
    lsi = torch.tensor([0])
    loc = torch.zeros(1, 4, 2, 1, 4, 2)
    w = torch.zeros(1, 4, 2, 1, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        MSDA.ms_deform_attn_forward(v, sh, lsi, loc, w, 128)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        MSDA.ms_deform_attn_backward(v, sh, lsi, loc, w, torch.zeros(1, 4, 64), 128)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(v.transpose(1, 2), sh, lsi, loc, w, 128)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["t_double", "t_float", "t_grad30", "t_grad32", "t_grad64", "t_grad71", "t_grad1025", "edge"])
def test_positional_calls_match_the_reference_outputs(tag):
    """The reference's own ops/test.py cases (seed 3) + border cases; expected outputs produced by the reference
    (tests/golden/gen_golden.py -> msda_core.npz)."""
    MSDA = _module()
    z = np.load(os.path.join(G, "msda_core.npz"))
    dt = torch.float32 if tag == "t_float" else torch.float64
    value, loc, w, go = (torch.from_numpy(z[f"{tag}/{k}"]).to(dt).cuda().contiguous() for k in ("value", "loc", "w", "grad_out"))
    sh = torch.as_tensor(z[f"{tag}/shapes"].tolist(), dtype=torch.int64)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    sh, lsi = sh.cuda(), lsi.cuda()
    out = MSDA.ms_deform_attn_forward(value, sh, lsi, loc, w, 128)
    res = MSDA.ms_deform_attn_backward(value, sh, lsi, loc, w, go, 128)
    assert isinstance(res, (list, tuple)) and len(res) == 3  # std::vector<at::Tensor> of ms_deform_attn.h:46
    gv, gl, gw = res
    torch.cuda.synchronize()
    tol = dict(rtol=1e-5, atol=1e-8) if dt == torch.float32 else dict(rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out.cpu().numpy(), z[f"{tag}/out"], **tol)
    gtol = dict(rtol=1e-4, atol=1e-7) if dt == torch.float32 else dict(rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(gv.cpu().numpy(), z[f"{tag}/grad_value"], **gtol)
    np.testing.assert_allclose(gw.cpu().numpy(), z[f"{tag}/grad_w"], **gtol)
    if tag != "edge":  # d/dloc is discontinuous exactly on pixel borders
        np.testing.assert_allclose(gl.cpu().numpy(), z[f"{tag}/grad_loc"], **gtol)


@pytest.mark.gpu
def test_reference_autograd_wrapper_shape_runs_on_the_stub():
    """A Function written exactly like the reference's MSDeformAttnFunction (ms_deform_attn_func.py:32-50: forward saves the five
    tensors, backward returns MSDA.ms_deform_attn_backward(...) as grad_value, None, None, grad_loc, grad_w, None) works on the
    stub and agrees with the product's own wrapper bit for bit in the forward pass."""
    MSDA = _module()
    from combo_avs_amd import msda

    class RefShaped(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, w, im2col_step):
            ctx.im2col_step = im2col_step
            out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, w, ctx.im2col_step)
            ctx.save_for_backward(value, shapes, lsi, loc, w)
            return out

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, grad_output):
            value, shapes, lsi, loc, w = ctx.saved_tensors
            gv, gl, gw = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, w, grad_output.contiguous(), ctx.im2col_step)
            return gv, None, None, gl, gw, None

    torch.manual_seed(5)
    shapes = [(7, 7), (14, 14), (28, 28)]
    B, S, M, D, L, P = 2, 1029, 8, 32, 3, 4
    sh = torch.as_tensor(shapes, dtype=torch.int64, device="cuda")
    lsi = torch.tensor([0, 49, 245], device="cuda")
    v = torch.randn(B, S, M, D, device="cuda")
    loc = torch.rand(B, S, M, L, P, 2, device="cuda") * 1.2 - 0.1
    w = torch.softmax(torch.randn(B, S, M, L * P, device="cuda"), -1).view(B, S, M, L, P)
    go = torch.randn(B, S, M * D, device="cuda")
    a = [t.clone().requires_grad_(True) for t in (v, loc, w)]
    b = [t.clone().requires_grad_(True) for t in (v, loc, w)]
    oa = RefShaped.apply(a[0], sh, lsi, a[1], a[2], 128)
    ob = msda.MSDeformAttnFunction.apply(b[0], sh, lsi, b[1], b[2], 128)
    assert torch.equal(oa, ob)
    ga = torch.autograd.grad(oa, a, go)
    gb = torch.autograd.grad(ob, b, go)
    # the stub's backward is the device-geometry pair of kernels, the product picks the one-launch windowed kernel: same gradients
    # up to the fixed-point grad_value accumulation (~1e-7 of max|grad_out|) and fp32 summation order
    for x, y, nm in zip(ga, gb, ("grad_value", "grad_loc", "grad_w")):
        torch.testing.assert_close(x, y, rtol=2e-4, atol=2e-5, msg=lambda m, nm=nm: nm + ": " + m)
