"""GPU parity of the ResNet backbones' stride-1 convolutions on the head's 3-product kernels (ops/convwrw.py: forward with the
fused bias / residual / ReLU epilogue, input gradients with the fused ReLU-gradient mask, tap-split 3x3 launches, grouped weight
images) against the library's fp32 convolution (detectron2 BottleneckBlock semantics)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def test_weight_images_equal_the_single_weight_presplit():
    """the grouped per-tap problems of ops.convwrw.weight_images write the same bits as combo_presplit_bf16x2_f32 of the
    explicitly permuted / flipped weight matrix"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import convwrw
    from combo_avs_amd.ops import linear as L
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(256, 64, 1, 1, generator=g).cuda(), torch.randn(64, 64, 3, 3, generator=g).cuda(),
          torch.randn(128, 256, 3, 3, generator=g).cuda(), torch.randn(64, 3, 7, 7, generator=g).cuda(),
          torch.randn(512, 256, 1, 1, generator=g).cuda()]
    geo = [((1, 1), (0, 0)), ((1, 1), (1, 1)), ((1, 1), (1, 1)), ((2, 2), (3, 3)), ((2, 2), (0, 0))]
    imgs = convwrw.weight_images(ws, geo)
    assert imgs[3] is None  # the 7x7 stem stays with the library
    # the stride-2 shortcut: forward image and - round 5: its input gradient is a GEMM over the output tokens - the transposed one
    assert torch.equal(imgs[4][0].view(torch.int32), L.presplit(ws[4].view(512, 256)).view(torch.int32))
    assert torch.equal(imgs[4][1].view(torch.int32), L.presplit(ws[4].view(512, 256).t().contiguous()).view(torch.int32))
    for w, im in zip(ws[:3], imgs[:3]):
        cout, cin, k, _ = w.shape
        fwd = L.presplit(w.permute(0, 2, 3, 1).reshape(cout, k * k * cin).contiguous())
        dx = L.presplit(w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, k * k * cout).contiguous())
        assert torch.equal(im[0].view(torch.int32), fwd.view(torch.int32))
        assert torch.equal(im[1].view(torch.int32), dx.view(torch.int32))


CASES = [  # B, H, cin, cout, k, residual
    (40, 56, 64, 256, 1, True),     # res2 conv3 + identity
    (40, 56, 256, 64, 1, False),    # res2 conv1 (64 output channels: skinny tiles)
    (40, 14, 1024, 256, 1, False),  # res4 conv1: split-K (few tiles, long K)
    (40, 7, 512, 2048, 1, True),    # res5 conv3
    (5, 56, 64, 64, 3, False),      # res2 conv2: 64 channels, library weight gradient
    (40, 14, 256, 256, 3, False),   # res4 conv2: tap split
    (40, 7, 512, 512, 3, False),    # res5 conv2: tap split
    (3, 28, 128, 128, 3, False),    # res3 conv2 at a ragged token count
    (10, 7, 512, 2048, 1, True),    # res5 conv3 at 10 frames: split-K with the residual added by the finishing sum
    (10, 14, 1024, 256, 1, False),  # res4 conv1 at 10 frames: split-K, bias + ReLU in the finishing sum
]


@pytest.mark.parametrize("B,H,cin,cout,k,with_res", CASES)
def test_conv_bias_act_on_the_3product_kernel_matches_the_library(B, H, cin, cout, k, with_res):
    """forward: relu(conv + bias (+ residual)) within 2e-5 of the output range of an fp64 evaluation (the library's fp32 kernel is
    held to the same bound); gradients (input, weight, residual) against the library path of the same module: relative L2 <= 1e-2.
    An output within round-off of zero may sit on the other side of the ReLU (~1e-5 of the entries): that moves one entry of the
    residual gradient, one ROW of the input gradient and a little of every weight-gradient entry, hence the entry-wise bounds:
    < 0.1 % (residual) / < 1 % (input) of the entries off by more than 1e-3 RMS, none for the weight gradient (measured: one
    flip among 2 352 tokens of a 3x3 layer = 9 input-gradient rows = 5.6e-3 relative L2, 0.38 % of the entries)."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import convwrw
    g = torch.Generator().manual_seed(B * 100 + cin + k)
    x = _cl(torch.randn(B, cin, H, H, generator=g).relu_().cuda())
    w = (torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5).cuda()
    b = torch.randn(cout, generator=g).cuda() * 0.3
    res = _cl(torch.randn(B, cout, H, H, generator=g).cuda()) if with_res else None
    probe = _cl(torch.randn(B, cout, H, H, generator=g).cuda())
    ref64 = F.conv2d(x.double(), w.double(), b.double(), 1, k // 2)
    if with_res:
        ref64 = ref64 + res.double()
    ref64 = ref64.relu_()
    out = {}
    for mode in ("own", "library"):
        xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        rr = res.clone().requires_grad_(True) if with_res else None
        images = convwrw.weight_images([ww.detach()], [((1, 1), (k // 2, k // 2))])[0] if mode == "own" else None
        assert (images is not None) == (mode == "own")
        prev = convwrw.DX_OWN
        convwrw.DX_OWN = 3 if mode == "own" else 0
        try:
            y = convwrw.conv_bias_act(xx * 1.0, ww, b, 1, k // 2, images, residual=rr)
            grads = torch.autograd.grad((y * probe).sum(), [xx, ww] + ([rr] if with_res else []))
        finally:
            convwrw.DX_OWN = prev
        out[mode] = (y.detach(), grads)
    scale = float(ref64.abs().max())
    for mode in out:
        assert float((out[mode][0].double() - ref64).abs().max()) <= 2e-5 * scale, mode
    for name, a, r in zip(("dx", "dw", "dres"), out["own"][1], out["library"][1]):
        rel = float((a - r).norm() / r.norm())
        rms = float(r.pow(2).mean().sqrt())
        frac = float(((a - r).abs() > 1e-3 * rms).float().mean())
        assert rel <= 1e-2 and frac < {"dx": 1e-2, "dw": 1.0, "dres": 1e-3}[name], (name, rel, frac)


def test_relu_gradient_mask_folded_into_the_3x3_input_gradient():
    """mask_dx: dX of the own 3x3 input-gradient kernel already multiplied by [x > 0] equals the unmasked result times the mask"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import convwrw
    g = torch.Generator().manual_seed(7)
    x = _cl(torch.randn(8, 256, 14, 14, generator=g).relu_().cuda())
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).cuda().requires_grad_(True)
    probe = _cl(torch.randn(8, 256, 14, 14, generator=g).cuda())
    images = convwrw.weight_images([w.detach()], [((1, 1), (1, 1))])[0]
    got = []
    for mask_dx in (False, True):
        xx = x.clone().requires_grad_(True)
        y = convwrw._ConvWrw.apply(xx, w, 3, mask_dx, images)
        got.append(torch.autograd.grad((y * probe).sum(), xx)[0])
    assert torch.equal(got[1], got[0] * (x > 0))


def test_resnet_features_and_gradients_own_forward_vs_library():
    """the whole fp32 ResNet-50 at 8 frames: res2..res5 features of the 3-product forward within 1e-4 of the feature range of the
    library forward (50 layers deep), parameter gradients within 5 % relative L2 per tensor (measured: up to 2.2 %) (ReLU sign flips at round-off level
    move individual entries; the sums they feed stay)"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.backbone import ResNet
    from combo_avs_amd.ops import convwrw
    torch.manual_seed(0)
    net = ResNet(50).cuda()
    for m in net.modules():
        if hasattr(m, "running_var"):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.uniform_(0.8, 1.2)
            m.bias.normal_(0, 0.1)
    net._drop_constants()
    x = torch.randn(8, 3, 224, 224, device="cuda")
    res = {}
    for mode in (True, False):
        prev = convwrw.FWD_X3, convwrw.DX_OWN
        convwrw.FWD_X3, convwrw.DX_OWN = mode, (3 if mode else 2)
        try:
            feats = net(x)
            loss = sum((f * torch.sin(torch.arange(f.numel(), device="cuda").view_as(f) * 0.37)).sum() for f in feats.values())
            params = [p for p in net.parameters() if p.requires_grad]
            grads = torch.autograd.grad(loss, params)
        finally:
            convwrw.FWD_X3, convwrw.DX_OWN = prev
        res[mode] = ({k: v.detach() for k, v in feats.items()}, grads)
    for k in res[True][0]:
        a, r = res[True][0][k], res[False][0][k]
        assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()), k
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    for n, a, r in zip(names, res[True][1], res[False][1]):
        rel = float((a - r).norm() / r.norm().clamp_min(1e-30))
        assert rel <= 5e-2, (n, rel)


@pytest.mark.parametrize("B,H,W,cin,cout,k", [(40, 28, 28, 256, 256, 3), (40, 14, 14, 1024, 2048, 1), (3, 15, 9, 128, 128, 3),
                                                (2, 7, 5, 256, 512, 1), (10, 56, 56, 128, 128, 3)])
def test_stride2_forward_on_the_own_kernel(B, H, W, cin, cout, k):
    """the first block of res3 / res4 / res5: 3x3 / stride 2 / pad 1 and the 1x1 / stride 2 shortcut (odd map sizes included),
    forward + bias + ReLU within 2e-5 of the output range of an fp64 evaluation; bit-identical from call to call; the backward
    pass (round 5: weight gradients and the 1x1 input gradient on own kernels, the 3x3 input gradient the library's) against
    autograd of F.conv2d"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import convwrw
    g = torch.Generator().manual_seed(H * 10 + k)
    x = _cl(torch.randn(B, cin, H, W, generator=g).cuda())
    w = (torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5).cuda().requires_grad_(True)
    b = torch.randn(cout, generator=g).cuda() * 0.3
    geo = ((2, 2), (k // 2, k // 2))
    assert convwrw.weight_kind(w, *geo) == 20 + k
    images = convwrw.weight_images([w.detach()], [geo])[0]
    assert (images[1] is None) == (k == 3)  # the 3x3 stride-2 input gradient stays the library's; the 1x1 one is own (round 5)
    xx = x.clone().requires_grad_(True)
    y = convwrw.conv_bias_act(xx, w, b, 2, k // 2, images)
    ref = F.conv2d(x.double(), w.detach().double(), b.double(), 2, k // 2).relu_()
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    y2 = convwrw.conv_bias_act(x.clone().requires_grad_(True), w, b, 2, k // 2, images)
    assert torch.equal(y, y2)
    probe = torch.randn(y.shape, generator=g).cuda()
    dx, dw = torch.autograd.grad((y * probe).sum(), (xx, w))
    x3 = x.clone().requires_grad_(True)
    dx_ref, dw_ref = torch.autograd.grad((F.conv2d(x3, w, b, 2, k // 2).relu() * probe).sum(), (x3, w))
    for a, r in ((dx, dx_ref), (dw, dw_ref)):
        assert float((a - r).norm() / r.norm()) <= 1e-2


def test_vggish_forward_on_the_own_kernels_matches_the_library_and_repeats_bit_for_bit():
    """audio_backbone/torchvggish/vggish.py:9-27 of the reference: the no-grad forward with the five wide convolutions on the own
    kernel against nn.Sequential (the library) within 1e-5 of the embedding range; two calls give identical bits"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.backbone import VGGish
    torch.manual_seed(0)
    net = VGGish().cuda().eval()
    x = torch.randn(40, 1, 96, 64, device="cuda")
    with torch.no_grad():
        a = net(x)
        b = net(x)
        ref = net.embeddings(net.features(x).permute(0, 2, 3, 1).reshape(40, -1))
    assert torch.equal(a, b)
    assert float((a - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_three_way_fanout_sums_all_consumers_in_the_relu_gradient_pass():
    """ops.biasact.bias_act(fanout=3): a stage's last block hands its output to the next stage's first convolution, its shortcut
    and the head as three aliases; dx = (dy1 + dy2 + dy3) . [y > 0] in one pass (combo_relu_grad3_f32) equals autograd on the
    plain expression"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.biasact import bias_act
    g = torch.Generator().manual_seed(11)
    z = _cl(torch.randn(4, 64, 9, 7, generator=g).cuda())
    b = torch.randn(64, generator=g).cuda()
    res = _cl(torch.randn(4, 64, 9, 7, generator=g).cuda())
    w = [_cl(torch.randn(4, 64, 9, 7, generator=g).cuda()) for _ in range(3)]
    zz, rr = z.clone().requires_grad_(True), res.clone().requires_grad_(True)
    a, b2, c = bias_act(zz * 1.0, b, rr, fanout=3)
    assert a.data_ptr() == b2.data_ptr() == c.data_ptr()
    got = torch.autograd.grad((a * w[0]).sum() + (b2 * w[1]).sum() + (c * w[2]).sum(), (zz, rr))
    z2, r2 = z.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y = torch.relu(z2 + b[None, :, None, None] + r2)
    ref = torch.autograd.grad((y * (w[0] + w[1] + w[2])).sum(), (z2, r2))
    for x, r in zip(got, ref):
        torch.testing.assert_close(x, r, rtol=1e-6, atol=1e-6)
    # two of three consumers without a gradient: still the masked sum of what arrives
    zz = z.clone().requires_grad_(True)
    a, b2, c = bias_act(zz * 1.0, b, None, fanout=3)
    got = torch.autograd.grad((b2 * w[1]).sum(), zz)[0]
    z2 = z.clone().requires_grad_(True)
    ref = torch.autograd.grad((torch.relu(z2 + b[None, :, None, None]) * w[1]).sum(), z2)[0]
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("ks,stride", [(3, 1), (3, 2), (1, 1), (1, 2)])
@pytest.mark.parametrize("cin,cout", [(64, 64), (80, 132), (128, 50)])
@pytest.mark.parametrize("aux_mode", [0, 1, 2])
def test_conv_entry_point_through_the_c_abi_every_epilogue_and_split(ks, stride, cin, cout, aux_mode):
    """combo_conv_nhwc_x3_epi_f32 called directly: 3x3 / 1x1, stride 1 / 2, odd map sizes, channel counts that are not multiples
    of the tile (cout = 50: the scalar-store path), bias + ReLU, aux as residual (1) or as ReLU-gradient mask (2), un-split and
    every legal tap split - against an fp64 evaluation (2e-5 of the output range)"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    from combo_avs_amd.ops import linear as L
    g = torch.Generator().manual_seed(ks * 100 + stride * 10 + cin + aux_mode)
    B, H, W = 3, 9, 7
    ho, wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    x = torch.randn(B, H, W, cin, generator=g).cuda()                 # NHWC tokens
    w = (torch.randn(cout, cin, ks, ks, generator=g) * (cin * ks * ks) ** -0.5).cuda()
    b = torch.randn(cout, generator=g).cuda()
    aux = torch.randn(B * ho * wo, cout, generator=g).cuda()
    img = L.presplit(w.permute(0, 2, 3, 1).reshape(cout, ks * ks * cin).contiguous())
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride, ks // 2).permute(0, 2, 3, 1).reshape(-1, cout)
    if aux_mode == 1:
        ref = ref + aux.double()
    ref = ref.relu()
    if aux_mode == 2:
        ref = ref * (aux.double() > 0)
    lib, st = _lib.lib(), _lib.current_stream()
    M = B * ho * wo
    for splits in ((1, 3, 9) if ks == 3 and cin % 32 == 0 and cout % 4 == 0 else (1,)):
        y = torch.full((M, cout), float("nan"), device="cuda")
        ws = torch.empty(splits, M, cout, device="cuda") if splits > 1 else None
        rc = lib.combo_conv_nhwc_x3_epi_f32(x.data_ptr(), cin, img.data_ptr(), b.data_ptr(), aux.data_ptr() if aux_mode else None, aux_mode,
                                            y.data_ptr(), cout, B, H, W, cin, cout, ks, stride, 1, splits, _lib.ptr(ws), st)
        assert rc == 0, (rc, splits)
        err = float((y.double() - ref).abs().max())
        assert err <= 2e-5 * float(ref.abs().max()) + 1e-6, (splits, err)


def test_mask_logits_of_the_whole_model_stay_within_the_north_star_bound_with_the_3product_backbones():
    """BASELINE.json north_star: mask logits within 1e-3 rel.  The whole COMBO-R50 model (2 clips x 5 frames, training forward) with
    the backbones' convolutions on the 3-product kernel against the same model with the library's fp32 convolutions: every mask
    logit of all 10 prediction heads within 1e-3 RMS(head) + 1e-3 |ref| up to a 0.1 % budget per head for re-routed queries
    (measured: 0 of 3.1 M logits per head, max deviation 3e-5 .. 2.3e-4 RMS; tools/probe_x3_forward_logits.py)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import graph_compare as GC
    from combo_avs_amd.ops import convwrw
    model, opt, batches, _ = GC.build("r50")
    grabbed = {}
    h = model.sem_seg_head.predictor.register_forward_hook(lambda m, i, o: grabbed.__setitem__("x", o["_logits_all"].detach().clone()))
    prev = convwrw.FWD_X3
    try:
        out = {}
        for mode in (False, True):
            convwrw.FWD_X3 = mode
            model(batches[0])
            out[mode] = grabbed["x"]
    finally:
        convwrw.FWD_X3 = prev
        h.remove()
    ref, got = out[False], out[True]
    assert ref.shape[0] == 10 and not torch.equal(ref, got)
    for head in range(10):
        rms = ref[head].pow(2).mean().sqrt()
        share = float(((got[head] - ref[head]).abs() > 1e-3 * rms + 1e-3 * ref[head].abs()).float().mean())
        assert share <= 1e-3, (head, share)


@pytest.mark.parametrize("B,H,W,cin,cout,ksize,stride", [
    (2, 56, 56, 64, 64, 3, 1),      # res2's 64-channel 3x3 layers: a 128-column K tile spans two taps, K = 576 is ragged
    (3, 28, 28, 128, 128, 3, 2),    # res3.0 conv2 (stride 2)
    (6, 15, 13, 256, 256, 3, 2),    # odd map: output ceil(H / 2) x ceil(W / 2), taps leaving the map on every side
    (2, 28, 28, 256, 512, 1, 2),    # res3.0 shortcut (1x1, stride 2)
    (2, 14, 14, 1024, 2048, 1, 2),  # res5.0 shortcut
    (3, 9, 11, 64, 192, 3, 1),
])
def test_generalised_weight_gradient_kernel_vs_fp64(B, H, W, cin, cout, ksize, stride):
    """combo_conv_wgrad_x3_f32 (round 5: any Cin % 4 == 0, kernel size 1 / 3, stride 1 / 2) against torch's convolution weight
    gradient evaluated in fp64: the ResNet-50 weight gradients that were the library's until round 4."""
    from combo_avs_amd.ops import conv3x3 as C3
    torch.manual_seed(B + H + cin + ksize + stride)
    pad = ksize // 2
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, ksize, ksize, device="cuda") * 0.05
    Ho, Wo = -(-H // stride), -(-W // stride)
    dy = torch.randn(B, cout, Ho, Wo, device="cuda").contiguous(memory_format=torch.channels_last)
    dw = C3._wgrad_tokens(C3._tokens(dy), C3._tokens(x), B, H, W, cin, cout, ksize=ksize, stride=stride)
    torch.cuda.synchronize()
    ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, (stride, stride), (pad, pad), (1, 1), False, (0, 0), 1,
                                              (False, True, False))[1]
    assert dw.shape == ref.shape
    err = (dw.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * scale, (err, scale)  # 3-product bf16 split: ~2^-17 per product, long reductions average it down


def test_resnet_weight_gradients_own_kernels_vs_library_for_every_layer():
    """every convolution weight gradient of the fp32 ResNet-50 (and the gradient of its input) with the round-5 switches on (64-channel
    and stride-2 weight gradients, the 1x1 stride-2 input gradients on the own kernels) against the same backward pass with them
    off (the library's kernels for those layers)"""
    from combo_avs_amd.backbone import ResNet
    from combo_avs_amd.ops import convwrw
    torch.manual_seed(0)
    net = ResNet(50).cuda().train()
    x = torch.randn(2, 3, 96, 96, device="cuda", requires_grad=True)  # (its gradient passes through every input-gradient kernel)
    res = {}
    for mode in (True, False):
        prev = convwrw.WGRAD_ANY_C, convwrw.WGRAD_S2, convwrw.DX_S2_1X1
        convwrw.WGRAD_ANY_C = convwrw.WGRAD_S2 = convwrw.DX_S2_1X1 = mode
        try:
            feats = net(x)
            loss = sum(f.float().pow(2).mean() for f in feats.values())
            names = [n for n, p in net.named_parameters() if p.requires_grad and n.endswith("weight") and p.dim() == 4]
            params = [dict(net.named_parameters())[n] for n in names]
            res[mode] = dict(zip(names + ["input"], torch.autograd.grad(loss, params + [x])))
        finally:
            convwrw.WGRAD_ANY_C, convwrw.WGRAD_S2, convwrw.DX_S2_1X1 = prev
    worst = []
    for n in res[True]:
        a, b = res[True][n].double(), res[False][n].double()
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        if rel > 2e-4:
            worst.append((n, rel))
    assert not worst, worst


@pytest.mark.parametrize("B,H,W,C", [(2, 28, 28, 256), (3, 15, 13, 64), (1, 7, 7, 1024)])
def test_expand_stride2_places_the_compact_gradient_on_the_even_pixels(B, H, W, C):
    from combo_avs_amd import _lib
    Ho, Wo = -(-H // 2), -(-W // 2)
    src = torch.randn(B, Ho, Wo, C, device="cuda")
    dst = torch.full((B, H, W, C), float("nan"), device="cuda")
    _lib.check(_lib.lib().combo_expand_stride2_f32(src.data_ptr(), dst.data_ptr(), B, H, W, C, _lib.current_stream()), "combo_expand_stride2_f32")
    ref = torch.zeros(B, H, W, C, device="cuda")
    ref[:, ::2, ::2] = src
    assert torch.equal(dst, ref)
