"""GPU parity of the implicit-GEMM 3x3 convolution (ops/conv3x3.py on csrc/gemm_f32.hip forward, csrc/gemm_nt3.hip dX, csrc/gemm_tn.hip dW; CONV = true)
against a float64 CPU convolution: forward, input gradient, weight gradient, bias gradient; square / non-square / tiny
maps, ragged token tiles, and the production shape of the FPN output layer (reference: pixel_decoder/msdeformattn.py:281-286)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()


@pytest.mark.parametrize("B,cin,cout,H,W,bias", [(2, 128, 128, 14, 14, True), (3, 256, 128, 7, 9, False), (1, 128, 256, 2, 2, True),
                                                  (5, 128, 128, 17, 5, False), (4, 256, 256, 56, 56, False)])
def test_conv3x3_forward_backward_vs_fp64(B, cin, cout, H, W, bias):
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import conv3x3 as C
    torch.manual_seed(B * 1000 + H)
    x = torch.randn(B, cin, H, W).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    b = torch.randn(cout) if bias else None
    g = torch.randn(B, cout, H, W).contiguous(memory_format=torch.channels_last)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if bias else None
    yd = F.conv2d(xd, wd, bd, padding=1)
    grads_d = torch.autograd.grad(yd, (xd, wd) + ((bd,) if bias else ()), g.double())

    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=bias).cuda()
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert C.usable(conv, xg)
    wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    y = C.conv3x3(xg, wg, bg)
    assert y.shape == (B, cout, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    grads = torch.autograd.grad(y, (xg, wg) + ((bg,) if bias else ()), g.cuda().contiguous(memory_format=torch.channels_last))
    assert rel_err(y, yd) < 1e-6, rel_err(y, yd)  # forward: fp32 grade (fp16-piece products by default, exact fp32 MFMA with COMBO_HEAD_FORWARD=fp32)
    for got, ref, name in zip(grads, grads_d, ("dx", "dw", "db")):
        assert got.shape == ref.shape, name
        assert rel_err(got, ref) < 2e-5, (name, rel_err(got, ref))


def test_conv2d_wrapper_routes_3x3_to_the_hip_kernels():
    """modeling.layers.Conv2d (the detectron2 wrapper equivalent) must take the HIP path for the FPN output layer and give
    the library's result."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    from combo_avs_amd.modeling.layers import Conv2d
    torch.manual_seed(1)
    conv = Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False).cuda()
    x = torch.randn(2, 256, 28, 28, device="cuda").contiguous(memory_format=torch.channels_last)
    _lib.start_timing()
    y = conv(x)
    timed = _lib.stop_timing()
    # the launch went through the package's implicit GEMM - csrc/gemm_nt3.hip (CONV, the default 3 x fp16-piece forward mode) or
    # csrc/gemm_f32.hip (CONV, `--head-dtype fp32`) - not MIOpen
    assert len(timed.get("conv3x3_f32", [])) + len(timed.get("conv3x3_bf16", [])) == 1, timed
    ref = F.conv2d(x.double().cpu(), conv.weight.detach().double().cpu(), padding=1)
    assert rel_err(y, ref) < 1e-6  # forward: fp32 grade (either forward mode)
