"""GPU parity tests of the MSDeformAttn core op: HIP kernels (through the C ABI)
vs the committed golden vectors from the reference and vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import combo_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def dev(t):
    return t.cuda().contiguous()


def run_hip(value, shapes, loc, w, grad_out=None, algo=0):
    from combo_avs_amd import msda
    msda.set_algo(algo)
    try:
        sh = torch.as_tensor(shapes, dtype=torch.int64)
        lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
        v, l, a = dev(value).requires_grad_(True), dev(loc).requires_grad_(True), dev(w).requires_grad_(True)
        out = msda.MSDeformAttnFunction.apply(v, sh.cuda(), lsi.cuda(), l, a, 128)
        res = [out.detach().cpu()]
        if grad_out is not None:
            gv, gl, gw = torch.autograd.grad(out, (v, l, a), dev(grad_out))
            res += [gv.cpu(), gl.cpu(), gw.cpu()]
        torch.cuda.synchronize()
        return res
    finally:
        msda.set_algo(0)


@pytest.mark.parametrize("tag", ["t_double", "t_float", "t_grad30", "t_grad32", "t_grad64", "t_grad71", "t_grad1025", "edge"])
def test_reference_unit_cases(tag):
    """The reference's own test cases (ops/test.py, seed 3) + border cases, outputs produced by the reference."""
    z = np.load(os.path.join(G, "msda_core.npz"))
    dt = torch.float32 if tag == "t_float" else torch.float64
    value, loc, w = (torch.from_numpy(z[f"{tag}/{k}"]).to(dt) for k in ("value", "loc", "w"))
    go = torch.from_numpy(z[f"{tag}/grad_out"]).to(dt)
    out, gv, gl, gw = run_hip(value, z[f"{tag}/shapes"].tolist(), loc, w, go)
    tol = dict(rtol=1e-5, atol=1e-8) if dt == torch.float32 else dict(rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out.numpy(), z[f"{tag}/out"], **tol)
    gtol = dict(rtol=1e-4, atol=1e-7) if dt == torch.float32 else dict(rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(gv.numpy(), z[f"{tag}/grad_value"], **gtol)
    np.testing.assert_allclose(gw.numpy(), z[f"{tag}/grad_w"], **gtol)
    if tag != "edge":  # d/dloc is discontinuous exactly on pixel borders
        np.testing.assert_allclose(gl.numpy(), z[f"{tag}/grad_loc"], **gtol)


def prod_inputs(B=2, shapes=((7, 7), (14, 14), (28, 28)), M=8, D=32, P=4, seed_tag="prod"):
    S = sum(h * w for h, w in shapes)
    L = len(shapes)
    v = synth.synth_tensor(f"{seed_tag}.value", (B, S, M, D), 0)
    refp = synth.synth_tensor(f"{seed_tag}.ref", (B, S, 1, 1, 1, 2), 0, kind="unit")
    off = synth.synth_tensor(f"{seed_tag}.off", (B, S, M, L, P, 2), 0, scale=2.5)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = refp + off / norm[None, None, None, :, None, :]
    w = torch.softmax(synth.synth_tensor(f"{seed_tag}.w", (B, S, M, L * P), 0), -1).view(B, S, M, L, P)
    return v, list(shapes), loc, w


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_production_shape_vs_reference_digests(algo):
    """R50-S4 @224 shape (S=Lq=1029, M=8, D=32, L=3, P=4): both kernel families against the reference's
    grid_sample core (digests in msda_core.npz) to fp32 round-off."""
    z = np.load(os.path.join(G, "msda_core.npz"))
    v, shapes, loc, w = prod_inputs()
    go = synth.synth_tensor("prod.grad_out", (2, 1029, 256), 0)
    out, gv, gl, gw = run_hip(v, shapes, loc, w, go, algo=algo)
    for nm, t in (("out", out), ("grad_value", gv), ("grad_loc", gl), ("grad_w", gw)):
        synth.check_digest(t, synth.unpack(f"prod/{nm}", z), f"prod/{nm}", rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_production_shape_vs_oracle_full_tensor(algo):
    v, shapes, loc, w = prod_inputs(B=3, seed_tag="prod3")
    go = synth.synth_tensor("prod3.grad_out", (3, 1029, 256), 0)
    v.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    ref = O.ms_deform_attn_core(v, shapes, loc, w)
    rgv, rgl, rgw = torch.autograd.grad(ref, (v, loc, w), go)
    out, gv, gl, gw = run_hip(v.detach(), shapes, loc.detach(), w.detach(), go, algo=algo)
    torch.testing.assert_close(out, ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gv, rgv, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gw, rgw, rtol=1e-4, atol=1e-5)
    # grad_loc: exclude samples within 1e-4 px of a pixel border (derivative jumps there)
    torch.testing.assert_close(gl, rgl, rtol=1e-3, atol=2e-4)


def test_ragged_and_tiny_shapes():
    """Lq != S, non-square levels, one level, P=1, B=1; D=16/64 (generic vec4 path) and D=6 (scalar path)."""
    for (shapes, M, D, P, Lq) in [(((3, 5),), 1, 16, 1, 7), (((4, 4), (2, 9)), 3, 64, 2, 33), (((5, 3), (1, 1)), 2, 6, 3, 5)]:
        S = sum(h * w for h, w in shapes)
        L = len(shapes)
        tag = f"rag{S}_{M}_{D}"
        v = synth.synth_tensor(tag + ".v", (2, S, M, D), 0)
        loc = synth.synth_tensor(tag + ".loc", (2, Lq, M, L, P, 2), 0, kind="unit") * 1.4 - 0.2
        w = synth.synth_tensor(tag + ".w", (2, Lq, M, L, P), 0, kind="unit")
        go = synth.synth_tensor(tag + ".go", (2, Lq, M * D), 0)
        v.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
        ref = O.ms_deform_attn_core(v, list(shapes), loc, w)
        rg = torch.autograd.grad(ref, (v, loc, w), go)
        out, gv, gl, gw = run_hip(v.detach(), list(shapes), loc.detach(), w.detach(), go)
        torch.testing.assert_close(out, ref.detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(gv, rg[0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(gl, rg[1], rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(gw, rg[2], rtol=1e-4, atol=1e-5)


def test_lds_and_generic_agree_bitwise_forward():
    """Both forward kernels accumulate the 48 taps in the same order -> identical bits."""
    v, shapes, loc, w = prod_inputs(B=2, seed_tag="bw")
    a = run_hip(v, shapes, loc, w, algo=1)[0]
    b = run_hip(v, shapes, loc, w, algo=2)[0]
    assert (a - b).abs().max().item() <= 1e-6
    c = run_hip(v, shapes, loc, w, algo=3)[0]  # tap-parallel kernel: per-tap partial sums, then combined
    assert (a - c).abs().max().item() <= 2e-5 * a.abs().max().item()


def test_full_size_properties_bs8():
    """BASELINE config 2 size (BT=40): size-independent properties instead of a CPU oracle run.
    (1) linearity in value, (2) constant value field + all-in-range samples -> out = const * sum(w),
    (3) sum over value-gradient equals sum of grad_out-weighted in-range tap weights (adjointness):
        <out, g> == <value, grad_value>."""
    B = 40
    v, shapes, loc, w = prod_inputs(B=B, seed_tag="full")
    v2 = synth.synth_tensor("full.v2", tuple(v.shape), 0)
    g = synth.synth_tensor("full.g", (B, 1029, 256), 0)
    o1, gv, _, _ = run_hip(v, shapes, loc, w, g)
    o2 = run_hip(v2, shapes, loc, w)[0]
    o12 = run_hip(2.0 * v + 3.0 * v2, shapes, loc, w)[0]
    torch.testing.assert_close(o12, 2.0 * o1 + 3.0 * o2, rtol=1e-4, atol=1e-4)
    # adjoint identity
    lhs = (o1.double() * g.double()).sum().item()
    rhs = (v.double() * gv.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-2
    # constant field, samples strictly inside every level
    loc_in = 0.3 + 0.4 * synth.synth_tensor("full.locin", tuple(loc.shape), 0, kind="unit")
    oc = run_hip(torch.full_like(v, 1.5), shapes, loc_in, w)[0]
    expect = (1.5 * w.sum((-1, -2)))[..., None].expand(-1, -1, -1, 32).reshape(B, 1029, 256)
    torch.testing.assert_close(oc, expect, rtol=1e-5, atol=1e-5)


def test_large_level_generic_path_512():
    """PVT@512-like encoder shape (S=5376 > LDS capacity) goes down the generic path; B=1 vs oracle."""
    shapes = ((16, 16), (32, 32), (64, 64))
    v, shapes, loc, w = prod_inputs(B=1, shapes=shapes, seed_tag="big")
    go = synth.synth_tensor("big.go", (1, 5376, 256), 0)
    v.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    ref = O.ms_deform_attn_core(v, shapes, loc, w)
    rg = torch.autograd.grad(ref, (v, loc, w), go)
    out, gv, gl, gw = run_hip(v.detach(), shapes, loc.detach(), w.detach(), go)
    torch.testing.assert_close(out, ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gv, rg[0], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(gw, rg[2], rtol=1e-4, atol=2e-5)
    assert_grad_loc_close(gl, rg[1], loc.detach(), shapes)  # the gather-only loc / w kernel of the windowed backward


def assert_grad_loc_close(gl, ref, loc, shapes, rtol=1e-3, atol=2e-4, border=1e-4):
    """grad_sampling_loc against the oracle's; d/dloc jumps where a sample sits on a pixel border, so samples within
    `border` pixels of one are left out (counted: they must stay a negligible fraction)."""
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=loc.dtype)[None, None, None, :, None, :]
    px = loc * wh - 0.5
    near = ((px - px.round()).abs() < border).any(-1, keepdim=True).expand_as(gl)
    assert near.float().mean().item() < 1e-2
    bad = ((gl - ref).abs() > atol + rtol * ref.abs()) & ~near
    assert not bool(bad.any()), (int(bad.sum()), float(((gl - ref).abs() * (~near)).max()))


def test_512_shape_bs8_oracle_at_b2_and_properties_at_full_size():
    """BASELINE configs[3] encoder shape (PVTv2-B5 @512: levels 16^2 / 32^2 / 64^2, S = 5376) at B = 8: the first two frames
    against the CPU oracle (forward + all gradients), the full batch through size-independent properties (linearity in
    value, adjointness <out, g> == <value, grad_value>)."""
    shapes = ((16, 16), (32, 32), (64, 64))
    B = 8
    v, shapes, loc, w = prod_inputs(B=B, shapes=shapes, seed_tag="big8")
    g = synth.synth_tensor("big8.g", (B, 5376, 256), 0)
    out, gv, gl, gw = run_hip(v, shapes, loc, w, g)
    v2, l2, w2 = (t[:2].clone().requires_grad_(True) for t in (v, loc, w))
    ref = O.ms_deform_attn_core(v2, shapes, l2, w2)
    rg = torch.autograd.grad(ref, (v2, l2, w2), g[:2])
    torch.testing.assert_close(out[:2], ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gv[:2], rg[0], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(gw[:2], rg[2], rtol=1e-4, atol=2e-5)
    assert_grad_loc_close(gl[:2], rg[1], loc[:2], shapes)
    vb = synth.synth_tensor("big8.v2", tuple(v.shape), 0)
    o2 = run_hip(vb, shapes, loc, w)[0]
    o12 = run_hip(2.0 * v + 3.0 * vb, shapes, loc, w)[0]
    torch.testing.assert_close(o12, 2.0 * out + 3.0 * o2, rtol=1e-4, atol=1e-4)
    lhs = (out.double() * g.double()).sum().item()
    rhs = (v.double() * gv.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-2


def test_512_shape_bt80_full_size_properties_all_three_gradients():
    """BASELINE configs[3] at its size: 8 clips x 10 frames = BT 80, S = 5376 (the windowed fixed-point grad_value kernel +
    the gather-only loc / w kernel).  The op is bilinear in (value, attention weight), so for ANY second weight tensor w2 and
    value v2:   <out(v, loc, w2), g> == <w2, grad_w>   and   <out(v2, loc, w), g> == <v2, grad_value>   (exact adjoint
    identities of grad_w / grad_value at full size); grad_loc: central difference of <out, g> along a random direction of the
    sampling locations; frames 0 and 79 against the CPU oracle (forward + all three gradients)."""
    shapes = ((16, 16), (32, 32), (64, 64))
    B = 80
    v, shapes, loc, w = prod_inputs(B=B, shapes=shapes, seed_tag="big80")
    g = synth.synth_tensor("big80.g", (B, 5376, 256), 0)
    out, gv, gl, gw = run_hip(v, shapes, loc, w, g)
    assert all(bool(torch.isfinite(t).all()) for t in (out, gv, gl, gw))
    for f in (0, 79):
        v1, l1, w1 = (t[f:f + 1].clone().requires_grad_(True) for t in (v, loc, w))
        ref = O.ms_deform_attn_core(v1, shapes, l1, w1)
        rg = torch.autograd.grad(ref, (v1, l1, w1), g[f:f + 1])
        torch.testing.assert_close(out[f:f + 1], ref.detach(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(gv[f:f + 1], rg[0], rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(gw[f:f + 1], rg[2], rtol=1e-4, atol=2e-5)
        assert_grad_loc_close(gl[f:f + 1], rg[1], loc[f:f + 1], shapes)
    gd = g.double()
    v2 = synth.synth_tensor("big80.v2", tuple(v.shape), 0)
    w2 = synth.synth_tensor("big80.w2", tuple(w.shape), 0)
    lhs = (run_hip(v2, shapes, loc, w)[0].double() * gd).sum().item()
    rhs = (v2.double() * gv.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0) + 1e-2, (lhs, rhs)
    lhs = (run_hip(v, shapes, loc, w2)[0].double() * gd).sum().item()
    rhs = (w2.double() * gw.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0) + 1e-2, (lhs, rhs)
    # grad_loc: directional derivative, step 2e-3 pixels of the finest level (second-order error; border crossings are rare)
    d = synth.synth_tensor("big80.dir", tuple(loc.shape), 0)
    eps = 2e-3 / 64
    fp = (run_hip(v, shapes, loc + eps * d, w)[0].double() * gd).sum().item()
    fm = (run_hip(v, shapes, loc - eps * d, w)[0].double() * gd).sum().item()
    fd, an = (fp - fm) / (2 * eps), (d.double() * gl.double()).sum().item()
    assert abs(fd - an) <= 2e-2 * max(abs(fd), abs(an)) + 1.0, (fd, an)


@pytest.mark.parametrize("shapes,B", [(((7, 7), (14, 14), (28, 28)), 3), (((16, 16), (32, 32), (64, 64)), 2)])
def test_windowed_fused_backward_matches_oracle_and_is_bitwise_deterministic(shapes, B):
    """csrc/msda_bwd.hip (the default backward for D = 32, P = 4): the one-launch kernel - workgroup = (frame, head, band of image
    rows of one level), all 32 channels, 2 x 32-bit fixed point per 64-bit LDS word - against the CPU oracle (all three
    gradients), against the two-kernel path (msda.WINDOWED_BACKWARD = False) and against itself (bitwise deterministic)."""
    from combo_avs_amd import msda
    tag = f"win{len(shapes)}_{shapes[-1][0]}"
    v, shapes, loc, w = prod_inputs(B=B, shapes=shapes, seed_tag=tag)
    S = sum(h * w_ for h, w_ in shapes)
    go = synth.synth_tensor(tag + ".go", (B, S, 256), 0)
    old = msda.WINDOWED_BACKWARD
    try:
        msda.WINDOWED_BACKWARD = False
        two = run_hip(v, shapes, loc, w, go)
        msda.WINDOWED_BACKWARD = True
        a = run_hip(v, shapes, loc, w, go)
        b = run_hip(v, shapes, loc, w, go)
    finally:
        msda.WINDOWED_BACKWARD = old
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)
    for x, y in zip(a[1:], two[1:]):  # two implementations, two summation orders: round-off apart
        assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max()) + 1e-6
    v1, l1, w1 = (t.clone().requires_grad_(True) for t in (v, loc, w))
    ref = O.ms_deform_attn_core(v1, shapes, l1, w1)
    rg = torch.autograd.grad(ref, (v1, l1, w1), go)
    torch.testing.assert_close(a[1], rg[0], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(a[3], rg[2], rtol=1e-4, atol=2e-5)
    assert_grad_loc_close(a[2], rg[1], loc, shapes)


def test_error_behaviour():
    from combo_avs_amd import msda
    v, shapes, loc, w = prod_inputs(B=1)
    sh = torch.as_tensor(shapes, dtype=torch.int64).cuda()
    lsi = torch.tensor([0, 49, 245]).cuda()
    with pytest.raises(RuntimeError):  # non-contiguous (ms_deform_attn_cuda.cu:33-37)
        msda.ms_deform_attn_forward(v.cuda().transpose(1, 2), sh, lsi, loc.cuda(), w.cuda())
    with pytest.raises(RuntimeError):  # half precision is not supported (ms_deform_attn_cuda.cu:69)
        msda.ms_deform_attn_forward(v.cuda().half(), sh, lsi, loc.cuda().half(), w.cuda().half())
    with pytest.raises(RuntimeError):  # int32 shapes
        msda.ms_deform_attn_forward(v.cuda(), sh.int(), lsi, loc.cuda(), w.cuda())


def test_prologue_kernel_and_merged_projection_match_torch():
    """Row a5: linear_cat + msda_prep (one GEMM + one kernel) against the reference formulation (two Linear layers, view,
    division, broadcast add, softmax; ms_deform_attn.py:101-118), forward and all gradients."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import linear_cat
    from combo_avs_amd.ops.msdaprep import msda_prep
    torch.manual_seed(0)
    B, Lq, C, M, L, P = 3, 1029, 256, 8, 3, 4
    q = torch.randn(B, Lq, C, device="cuda", requires_grad=True)
    w1 = (torch.randn(M * L * P * 2, C, device="cuda") * 0.05).requires_grad_(True)
    b1 = torch.randn(M * L * P * 2, device="cuda").requires_grad_(True)
    w2 = (torch.randn(M * L * P, C, device="cuda") * 0.05).requires_grad_(True)
    b2 = torch.randn(M * L * P, device="cuda").requires_grad_(True)
    ref = torch.rand(1, Lq, L, 2, device="cuda").expand(B, -1, -1, -1)
    norm = torch.tensor([[7.0, 7.0], [14.0, 14.0], [28.0, 28.0]], device="cuda")
    loc, attn = msda_prep(linear_cat(q, w1, b1, w2, b2), ref, norm, M, L, P)
    off = torch.nn.functional.linear(q, w1, b1).view(B, Lq, M, L, P, 2)
    lg = torch.nn.functional.linear(q, w2, b2).view(B, Lq, M, L * P)
    attn_r = lg.softmax(-1).view(B, Lq, M, L, P)
    loc_r = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    assert (loc - loc_r).abs().max() < 2e-5 and (attn - attn_r).abs().max() < 2e-6
    g1, g2 = torch.randn_like(loc), torch.randn_like(attn)
    got = torch.autograd.grad([loc, attn], [q, w1, b1, w2, b2], [g1, g2])
    want = torch.autograd.grad([loc_r, attn_r], [q, w1, b1, w2, b2], [g1, g2])
    for a, b in zip(got, want):
        assert (a - b).abs().max() <= 2e-4 * b.abs().max() + 1e-6, float((a - b).abs().max() / b.abs().max())
