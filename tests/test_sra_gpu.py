"""GPU parity tests of the spatial-reduction attention kernels (csrc/sra_attention.hip; row f2: PVTv2's attention,
models/modeling/backbone/pvtv2.py:104-118) through the C ABI: forward, dq, dkv against an fp32 evaluation of the reference
formula on the same bf16 operands, at the shapes of PVTv2-B5's four stages at 224 x 224 (49 keys) and 512 x 512 (256 keys),
ragged sizes, and against the library attention the kernels replace."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def reference(q, kv, h, scale, dout=None):
    """pvtv2.py:104-118 in fp32 on the bf16 operands: attn = softmax(q k^T * scale); x = (attn @ v).transpose(1, 2).reshape(B, N, C)"""
    B, N, C = q.shape
    d = C // h
    q32 = q.float().detach().requires_grad_(True)
    kv32 = kv.float().detach().requires_grad_(True)
    qh = q32.view(B, N, h, d).transpose(1, 2)
    k, v = kv32.view(B, -1, 2, h, d).unbind(2)
    k, v = k.transpose(1, 2), v.transpose(1, 2)
    attn = (qh @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    if dout is None:
        return out.detach()
    gq, gkv = torch.autograd.grad(out, (q32, kv32), dout.float())
    return out.detach(), gq, gkv


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


SHAPES = [  # (B, N, heads, Nk): PVTv2-B5 stages at 224 x 224 and 512 x 512 (one or two frames), then ragged sizes
    (2, 3136, 1, 49), (2, 784, 2, 49), (2, 196, 5, 49), (3, 49, 8, 49),
    (1, 16384, 1, 256), (1, 4096, 2, 256), (2, 1024, 5, 256), (2, 256, 8, 256),
    (2, 1000, 2, 100), (1, 77, 3, 7), (2, 33, 1, 129),
]


@pytest.mark.parametrize("B,N,h,Nk", SHAPES)
def test_forward_and_backward_match_the_fp32_formula(B, N, h, Nk):
    from combo_avs_amd.ops import sra
    torch.manual_seed(B * 1000 + N + h + Nk)
    C, scale = 64 * h, 64 ** -0.5
    q = (torch.randn(B, N, C, device="cuda") * 1.5).to(torch.bfloat16)
    kv = (torch.randn(B, Nk, 2 * C, device="cuda") * 1.5).to(torch.bfloat16)
    dout = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
    assert sra.usable(q, kv, h)
    qa, kva = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    out = sra.sra_attention(qa, kva, h, scale)
    gq, gkv = torch.autograd.grad(out, (qa, kva), dout)
    torch.cuda.synchronize()
    ref, rq, rkv = reference(q, kv, h, scale, dout)
    # the library path on the same operands: the yardstick for what bf16 operands / bf16 P cost
    ql, kvl = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    k, v = kvl.view(B, -1, 2, h, 64).unbind(2)
    lib = torch.nn.functional.scaled_dot_product_attention(ql.view(B, N, h, 64).transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=scale)
    lib = lib.transpose(1, 2).reshape(B, N, C)
    lq, lkv = torch.autograd.grad(lib, (ql, kvl), dout)
    e_out, e_q, e_kv = rel_l2(out, ref), rel_l2(gq, rq), rel_l2(gkv, rkv)
    l_out, l_q, l_kv = rel_l2(lib, ref), rel_l2(lq, rq), rel_l2(lkv, rkv)
    print(f"[sra B={B} N={N} h={h} Nk={Nk}] rel L2 own / library: out {e_out:.2e} / {l_out:.2e}, dq {e_q:.2e} / {l_q:.2e}, dkv {e_kv:.2e} / {l_kv:.2e}")
    assert torch.isfinite(out.float()).all() and torch.isfinite(gq.float()).all() and torch.isfinite(gkv.float()).all()
    # bf16 results: 2^-9 relative rounding of the stored values + bf16 probabilities inside; stated bound 1e-2 relative L2, and never
    # more than 1.5 x the library's own error + 2e-3
    assert e_out <= 1e-2 and e_q <= 1.5e-2 and e_kv <= 1.5e-2, (e_out, e_q, e_kv)
    assert e_out <= 1.5 * l_out + 2e-3 and e_q <= 1.5 * l_q + 2e-3 and e_kv <= 1.5 * l_kv + 2e-3
    # element-wise: no entry further than 3e-2 of the tensor's max from the fp32 formula
    assert float((out.float() - ref).abs().max()) <= 3e-2 * float(ref.abs().max())
    assert float((gq.float() - rq).abs().max()) <= 3e-2 * float(rq.abs().max()) + 1e-3
    assert float((gkv.float() - rkv).abs().max()) <= 3e-2 * float(rkv.abs().max()) + 1e-3


def test_runs_are_bitwise_reproducible():
    from combo_avs_amd.ops import sra
    torch.manual_seed(3)
    B, N, h, Nk = 2, 4096, 2, 256
    q = torch.randn(B, N, 64 * h, device="cuda").to(torch.bfloat16)
    kv = torch.randn(B, Nk, 128 * h, device="cuda").to(torch.bfloat16)
    dout = torch.randn(B, N, 64 * h, device="cuda").to(torch.bfloat16)
    res = []
    for _ in range(2):
        qa, kva = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        out = sra.sra_attention(qa, kva, h, 0.125)
        res.append((out.detach().clone(),) + tuple(g.clone() for g in torch.autograd.grad(out, (qa, kva), dout)))
    torch.cuda.synchronize()
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_sharp_softmax_and_large_scores_stay_finite():
    """scores of +-60 (a one-hot softmax) and a zero gradient: exp2 of large negative arguments, no NaN"""
    from combo_avs_amd.ops import sra
    B, N, h, Nk = 1, 64, 1, 49
    q = torch.zeros(B, N, 64, device="cuda")
    q[..., 0] = 60.0
    kv = torch.zeros(B, Nk, 128, device="cuda")
    kv[:, 7, 0] = 8.0   # key 7 wins every query by 60 after scaling
    kv[:, :, 64:] = torch.arange(Nk, device="cuda").float()[None, :, None]
    q, kv = q.to(torch.bfloat16).requires_grad_(True), kv.to(torch.bfloat16).requires_grad_(True)
    out = sra.sra_attention(q, kv, h, 0.125)
    gq, gkv = torch.autograd.grad(out, (q, kv), torch.ones_like(out))
    assert torch.allclose(out.float(), torch.full_like(out.float(), 7.0), atol=1e-2)
    assert torch.isfinite(gq.float()).all() and torch.isfinite(gkv.float()).all()
    assert float(gkv.float()[0, 7, 64:].sum()) == pytest.approx(64.0 * N, rel=1e-2)  # every query's gradient lands on v[7]


def test_pvt_backbone_features_and_gradients_own_attention_vs_library():
    """the whole PVTv2-B5 (bf16 recipe) with the own attention against the same backbone on F.scaled_dot_product_attention"""
    from combo_avs_amd.backbone_pvt import DropPath, PyramidVisionTransformerV2
    from combo_avs_amd.ops import sra
    torch.manual_seed(0)
    bb = PyramidVisionTransformerV2(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(4, 4, 4, 4), qkv_bias=True,
                                    norm_eps=1e-6, depths=(3, 6, 40, 3), sr_ratios=(8, 4, 2, 1), drop_rate=0.0, drop_path_rate=0.1,
                                    out_features=("res2", "res3", "res4", "res5")).cuda().train()
    for m_ in bb.modules():
        if isinstance(m_, DropPath):
            m_.p = 0.0
    x = torch.randn(2, 3, 224, 224, device="cuda")
    params = [p for n, p in bb.named_parameters() if n.endswith(("attn.q.weight", "attn.kv.weight", "attn.proj.weight"))][:12]
    res = {}
    for mode in (True, False):
        prev, sra.ENABLED = sra.ENABLED, mode
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                feats = bb(x)
            loss = sum(f.float().pow(2).mean() for f in feats.values())
            res[mode] = ({k: v.float().detach() for k, v in feats.items()}, torch.autograd.grad(loss, params))
        finally:
            sra.ENABLED = prev
    for k in res[True][0]:
        assert rel_l2(res[True][0][k], res[False][0][k]) <= 3e-2, (k, rel_l2(res[True][0][k], res[False][0][k]))
    # two bf16 computations of a 52-block backbone against each other (the library's attention backward uses atomics: its own
    # gradients move from run to run): measured 0.05 - 0.082 relative L2 on the attention weights of the first blocks
    for a, b in zip(res[True][1], res[False][1]):
        assert rel_l2(a, b) <= 1.5e-1, rel_l2(a, b)


@pytest.mark.parametrize("B,H,W,C,sr", [(10, 56, 56, 64, 8), (4, 28, 28, 128, 4), (2, 32, 32, 320, 2), (2, 30, 27, 64, 4)])
def test_sr_patch_gemm_equals_the_strided_convolution(B, H, W, C, sr):
    """backbone_pvt._sr_patch_gemm (round 6 option: the spatial-reduction convolution, kernel = stride, as a gather + GEMM over
    non-overlapping patches - reproducible without cudnn.deterministic) against F.conv2d on the same bf16 operands: forward and both
    gradients within bf16 round-off of the fp32 evaluation (pvtv2.py:76, :106-108); ragged maps drop the same border rows / columns."""
    import torch.nn.functional as F
    from combo_avs_amd import backbone_pvt as BP
    torch.manual_seed(B * 131 + C)
    x = (0.5 * torch.randn(B, H * W, C, device="cuda")).bfloat16().requires_grad_(True)
    w = (torch.randn(C, C, sr, sr, device="cuda") / (C * sr * sr) ** 0.5).bfloat16().requires_grad_(True)
    a = BP._sr_patch_gemm(x, w, B, H, W, C, sr)
    ref = F.conv2d(x.float().view(B, H, W, C).permute(0, 3, 1, 2), w.float(), None, sr).flatten(2).transpose(1, 2)
    assert a.shape == ref.shape and a.dtype == torch.bfloat16
    g = torch.randn_like(ref)
    ga = torch.autograd.grad(a, (x, w), g.bfloat16())
    gr = torch.autograd.grad(ref, (x, w), g)
    for got, want in ((a, ref), (ga[0], gr[0]), (ga[1], gr[1])):
        err = (got.float() - want).norm() / want.norm()
        assert float(err) < 8e-3, float(err)  # bf16 operands / results: 2^-9 per element
