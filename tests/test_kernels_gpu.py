"""GPU parity of the small fused kernels (fused AdamW, matcher, losses, masks, norms, column sums, PVT glue)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_fused_adamw_matches_torch_optimizer():
    """FlatAdamW (one flat buffer, fused HIP clip+AdamW, reference grouping rules) vs torch.optim.AdamW +
    clip_grad_norm_ as train_net.py:147-226 of the reference builds it."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.trainer import FlatAdamW, param_groups
    torch.manual_seed(0)

    def make():
        torch.manual_seed(1)
        m = torch.nn.ModuleDict({
            "backbone": torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.LayerNorm(17)),
            "pre_sam_backbone": torch.nn.Linear(17, 9),
            "head": torch.nn.Sequential(torch.nn.Linear(9, 5), torch.nn.GroupNorm(1, 5)),
            "emb": torch.nn.Embedding(7, 5)}).cuda()
        return m

    def loss_fn(m, x):
        y = m["head"](m["pre_sam_backbone"](m["backbone"](x)))
        return (y * m["emb"].weight[:y.shape[0]].sum(0)).pow(2).sum() * 50

    a, b = make(), make()
    groups = [{"params": [p], "lr": lr, "weight_decay": wd} for p, _, lr, wd in param_groups(b, 1e-3, 0.05)]
    assert sorted({(g["lr"], g["weight_decay"]) for g in groups}) == [(1e-4, 0.0), (1e-4, 0.05), (1e-3, 0.0), (1e-3, 0.05)]
    ref = torch.optim.AdamW(groups, 1e-3)
    opt = FlatAdamW(a, base_lr=1e-3, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01)
    for it in range(5):
        x = torch.randn(6, 33, device="cuda")
        opt.backward(loss_fn(a, x))
        opt.all_reduce_grads()
        opt.step()
        ref.zero_grad()
        loss_fn(b, x).backward()
        torch.nn.utils.clip_grad_norm_([p for g in groups for p in g["params"]], 0.01)
        ref.step()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-7, msg=n)


def test_fused_matcher_cost_vs_oracle():
    """csrc/matcher.hip against the oracle's per-frame cost (matcher.py:93-131 restated) on random problems,
    incl. points outside [0,1] (zero padding) and a frame with fewer instances than Gmax."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.modeling.matcher import HungarianMatcher
    from oracle import combo_oracle as O
    torch.manual_seed(0)
    N, Q, K1, G, h, w, H, W, P = 6, 100, 3, 2, 56, 56, 224, 224, 12544
    logits = torch.randn(N, Q, K1)
    masks = torch.randn(N, Q, h, w) * 4
    gt = (torch.rand(N, G, H, W) > 0.6).float()
    labels = torch.randint(0, K1 - 1, (N, G))
    pts = torch.rand(N, P, 2) * 1.1 - 0.05
    m = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=P)
    got = m.batched_cost(logits.cuda(), masks.cuda(), labels.cuda(), gt.cuda(), pts.cuda()).cpu()
    for n in range(N):
        ref = O.matcher_cost(logits[n], masks[n], labels[n], gt[n], pts[n:n + 1])
        torch.testing.assert_close(got[n], ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sem_mix_forward_backward(dtype):
    """Row a1: channel_weighted_block + mix on csrc/semmix.hip vs the oracle (fp64 torch), channels-last bf16/fp32."""
    import json, os
    import numpy as np
    import synth
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.modeling.semmix import channel_weighted_block, sem_mix
    from oracle import combo_oracle as O
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sem_mix.npz"))
    spec = json.loads(str(z["sem256/spec"]))
    blk = channel_weighted_block(256)
    blk.load_state_dict({k: synth.synth_param("sem256." + k, s) for k, s in spec})
    blk = blk.cuda()
    f = synth.synth_tensor("sem256.f", (3, 256, 6, 5), 0)
    p = synth.synth_tensor("sem256.p", (3, 256, 6, 5), 0)
    if dtype == torch.float32:  # golden fixture produced by the reference's own module
        out = sem_mix({"res2": f.cuda()}, {"res2": p.cuda()}, [blk])["res2"]
        np.testing.assert_allclose(out.detach().cpu().numpy(), z["sem256/mixed"], rtol=1e-5, atol=1e-5)
    fq, pq = f.to(dtype).float(), p.to(dtype).float()  # the values the kernel actually sees
    P = {"m.0." + k: v.detach().cpu().double().requires_grad_(True) for k, v in blk.state_dict().items()}
    fr, pr = fq.double().requires_grad_(True), pq.double().requires_grad_(True)
    ref = O.sem_mix(P, "m.", {"res2": fr}, {"res2": pr})["res2"]
    g = synth.synth_tensor("sem256.g", tuple(ref.shape), 0)
    names = [k for k, _ in blk.named_parameters()]
    rf, rp, *r_params = torch.autograd.grad(ref, (fr, pr) + tuple(P["m.0." + k] for k in names), g.double())
    fg = f.cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    pg = p.cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = sem_mix({"res2": fg}, {"res2": pg}, [blk])["res2"]
    assert out.dtype == torch.float32
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-5, atol=1e-5)
    gf, gp, *g_params = torch.autograd.grad(out, (fg, pg) + tuple(blk.parameters()), g.cuda())
    tol = dict(rtol=1e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gf.cpu().double(), rf, **tol)
    torch.testing.assert_close(gp.cpu().double(), rp, **tol)
    # the gate's parameters: their gradients come out of the graph recorded INSIDE the fused pool + gate + mix node (ops/semmix.py)
    for k, a, r in zip(names, g_params, r_params):
        assert float((a.cpu().double() - r).norm() / r.norm()) < (2e-2 if dtype == torch.bfloat16 else 1e-4), k


def test_fused_inference_tail_vs_oracle():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.infer import semantic_inference
    from oracle import combo_oracle as O
    torch.manual_seed(0)
    for K in (2, 11):
        logits = torch.randn(3, 100, K + 1)
        masks = torch.randn(3, 100, 56, 56) * 3
        ref = O.semantic_inference(logits, masks, (224, 224))
        got = semantic_inference(logits.cuda(), masks.cuda(), (224, 224)).cpu()
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


def test_fused_mask_loss_vs_oracle():
    """csrc/maskloss.hip: radix-select importance sampling picks exactly the k most uncertain points (as a set),
    BCE / dice and their gradient match the oracle's loss_masks on the same points."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import maskloss
    from oracle import combo_oracle as O
    torch.manual_seed(0)
    nmaps, NM, h, w, H, W = 30, 7, 56, 56, 224, 224
    NS, NR, k = 37632, 3136, 9408
    masks = (torch.randn(nmaps, h, w) * 3)
    masks[3] = 0.0  # massive ties: every |logit| equal -> ties must be resolved deterministically by index
    gt = (torch.rand(5, H, W) > 0.5).float()
    midx = torch.tensor([4, 0, 29, 3, 17, 8, 11])
    gidx = torch.tensor([0, 4, 2, 2, 1, 3, 0])
    over = torch.rand(NM, NS, 2)
    extra = torch.rand(NM, NR, 2)
    coords = maskloss.uncertain_points(masks.cuda(), midx.cuda(), over.cuda(), extra.cuda(), k).cpu()
    assert coords.shape == (NM, k + NR, 2)
    torch.testing.assert_close(coords[:, k:], extra)
    for n in range(NM):
        x = O.point_sample(masks[midx[n]][None, None], over[n:n + 1])[0, 0].abs()
        ref_idx = torch.topk(-x, k)[1]
        thr = x[ref_idx].max()
        got = coords[n, :k]
        # every selected point has |x| <= threshold, and the multiset of selected |x| equals the k smallest
        gx = O.point_sample(masks[midx[n]][None, None], got[None])[0, 0].abs()
        assert gx.max() <= thr + 1e-6
        torch.testing.assert_close(gx.sort()[0], x[ref_idx].sort()[0], rtol=1e-5, atol=1e-6)
    # losses + gradient on the selected points
    mg = masks.cuda().requires_grad_(True)
    bce, dice = maskloss.mask_losses(mg, midx.cuda(), gt.cuda(), gidx.cuda(), coords.cuda())
    gb, gd = torch.rand(NM), torch.rand(NM)
    (bce * gb.cuda()).sum().add((dice * gd.cuda()).sum()).backward()
    mr = masks.clone().requires_grad_(True)
    src = mr[midx][:, None]
    tgt = gt[gidx][:, None]
    logits = O.point_sample(src, coords).squeeze(1)
    labels = O.point_sample(tgt, coords).squeeze(1)
    rb = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels, reduction="none").mean(1)
    sg = logits.sigmoid()
    rd = 1 - (2 * (sg * labels).sum(-1) + 1) / (sg.sum(-1) + labels.sum(-1) + 1)
    (rb * gb).sum().add((rd * gd).sum()).backward()
    torch.testing.assert_close(bce.detach().cpu(), rb.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dice.detach().cpu(), rd.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mg.grad.cpu(), mr.grad, rtol=1e-3, atol=1e-6)


def test_fused_cosine_loss_vs_oracle():
    """csrc/cosine.hip (stats + gradient) against the oracle's similarity_loss (criterion.py:208-231)."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import maskloss
    from oracle import combo_oracle as O
    torch.manual_seed(0)
    n9, bt, Q, HW, nf = 3, 10, 100, 3136, 5
    mid = torch.randn(n9, bt, Q, HW)
    mid[:, 1] = mid[:, 0] * 0.9 + 0.1 * torch.randn(n9, Q, HW)  # a nearly parallel pair
    xr = mid.clone().requires_grad_(True)
    ref = torch.stack([O.similarity_loss(xr[i], nf) for i in range(n9)])
    w = torch.tensor([1.0, 2.0, 3.0])
    (ref * w).sum().backward()
    xg = mid.cuda().requires_grad_(True)
    dot, nrm = maskloss.cosine_stats(xg.reshape(n9 * bt, -1), nf)
    dot, nrm = dot.view(n9, bt // nf, nf), nrm.view(n9, bt // nf, nf)
    cos = dot[..., :-1] / torch.sqrt((nrm[..., :-1] + 1e-12) * (nrm[..., 1:] + 1e-12))
    c = 1 - cos
    got = (c * torch.exp(-c)).sum((1, 2)) / (bt // nf) / (nf - 1)
    (got * w.cuda()).sum().backward()
    torch.testing.assert_close(got.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-9)


@pytest.mark.parametrize("G", [1, 2, 3, 5, 6])
def test_device_lsap_equals_scipy(G):
    """csrc/lsap.hip: exact optimum == scipy.optimize.linear_sum_assignment (total cost and, without ties, the pairs)."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.modeling.matcher import HungarianMatcher
    m = HungarianMatcher(1, 1, 1, 16)
    rng = np.random.default_rng(G)
    N, Q, Gpad = 40, 100, max(G, 2)
    cost = rng.standard_normal((N, Q, Gpad)).astype(np.float32)
    cost[0, :, :] = np.round(cost[0] * 2) / 2  # many exact ties
    gcount = np.full(N, G, dtype=np.int32)
    gcount[1] = max(1, G - 1)  # ragged
    got = m.solve_device(torch.from_numpy(cost).cuda(), torch.from_numpy(gcount).cuda()).cpu().numpy()
    for n in range(N):
        g = gcount[n]
        r, c = linear_sum_assignment(cost[n, :, :g])
        rows = got[n, :g]
        assert len(set(rows.tolist())) == g and (got[n, g:] == -1).all()
        tot = cost[n, rows, np.arange(g)].sum()
        assert abs(tot - cost[n, r, c].sum()) <= 1e-5
        if n > 1:  # continuous costs: unique optimum -> identical pairs
            assert dict(zip(c.tolist(), r.tolist())) == dict(zip(range(g), rows.tolist()))


def test_device_lsap_surfaces_non_finite_costs_and_too_many_targets():
    """scipy.optimize.linear_sum_assignment raises ValueError on NaN / Inf costs (matcher.py:133).  The device solver keeps
    returning valid, distinct rows (no out-of-range index can reach the mask-loss gathers) AND sets a status bit that
    HungarianMatcher.check_status turns into that ValueError."""
    import numpy as np
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.modeling.matcher import HungarianMatcher
    m = HungarianMatcher(1, 1, 1, 16)
    rng = np.random.default_rng(0)
    cost = rng.standard_normal((4, 100, 3)).astype(np.float32)
    gcount = torch.full((4,), 3, dtype=torch.int32).cuda()
    got = m.solve_device(torch.from_numpy(cost).cuda(), gcount)
    m.check_status()  # clean
    cost[2, 17, 1] = np.nan
    cost[3, :, 0] = np.inf
    got = m.solve_device(torch.from_numpy(cost).cuda(), gcount).cpu().numpy()
    assert ((got >= 0) & (got < 100)).all() and all(len(set(r.tolist())) == 3 for r in got)
    with pytest.raises(ValueError, match="invalid numeric entries"):
        m.check_status()
    m.check_status()  # the word is cleared once reported
    gcount[1] = 5  # more targets than the padded width
    m.solve_device(torch.from_numpy(np.nan_to_num(cost, posinf=1.0)).cuda(), gcount)
    with pytest.raises(ValueError, match="more than"):
        m.check_status()


def test_backbone_fused_epilogue_matches_unfused_bf16():
    """ResNet forward/backward with the folded weights + fused bias/residual/ReLU epilogue (csrc/biasact.hip) against
    the per-convolution torch path (conv + FrozenBN affine + relu) under the same bf16 autocast."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.backbone import ResNet
    torch.manual_seed(0)
    m = ResNet(50).cuda().train()
    for mod in m.modules():
        if hasattr(mod, "running_var"):
            mod.running_var.uniform_(0.5, 1.5); mod.running_mean.normal_(0, 0.1); mod.weight.uniform_(0.8, 1.2); mod.bias.normal_(0, 0.1)
    m._bn_cache.clear()
    x = torch.randn(4, 3, 96, 96, device="cuda")
    params = [p for p in m.parameters()]

    def unfused(x):
        x = x.contiguous(memory_format=torch.channels_last)
        x = torch.nn.functional.max_pool2d(torch.relu(m.stem.conv1(x)), 3, 2, 1)
        out = {}
        for name in ("res2", "res3", "res4", "res5"):
            for blk in getattr(m, name):
                x = blk(x)
            out[name] = x
        return out
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = m(x)
        b = unfused(x)
    for k in a:
        assert a[k].dtype == torch.bfloat16
        err = (a[k].float() - b[k].float()).abs().max() / b[k].float().abs().max()
        assert err < 0.05, (k, float(err))  # bf16 activations through 50 layers; the fused path rounds once per epilogue
    ga = torch.autograd.grad(sum(v.float().pow(2).mean() for v in a.values()), params)
    gb = torch.autograd.grad(sum(v.float().pow(2).mean() for v in b.values()), params)
    cos = [float(torch.nn.functional.cosine_similarity(u.flatten().float(), v.flatten().float(), dim=0)) for u, v in zip(ga, gb)]
    # bf16 backward through 50 layers: the stem weight (deepest gradient) is the noisiest, everything else agrees closely
    assert min(cos) > 0.95 and sorted(cos)[len(cos) // 2] > 0.995, (min(cos), sorted(cos)[len(cos) // 2])


def test_bias_act_kernels_exact():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.biasact import bias_act
    torch.manual_seed(1)
    y0 = torch.randn(3, 64, 10, 12, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(y0).contiguous(memory_format=torch.channels_last)
    b = torch.randn(64, device="cuda")
    for res in (None, r):
        y = y0.clone().requires_grad_(True)
        rr = None if res is None else res.clone().requires_grad_(True)
        out = bias_act(y * 1.0, b, rr)  # (y * 1.0: a non-leaf that may be written in place)
        ref = y0.float() + b[None, :, None, None] + (0 if res is None else res.float())
        ref = ref.clamp_min(0).to(torch.bfloat16)
        assert torch.equal(out, ref)
        g = torch.randn_like(out)
        grads = torch.autograd.grad(out, [y] + ([rr] if rr is not None else []), g)
        want = (g.float() * (ref.float() > 0)).to(torch.bfloat16)
        for gg in grads:
            assert torch.equal(gg, want)


@pytest.mark.parametrize("B,H,W,C", [(3, 14, 14, 1280), (2, 7, 9, 64), (4, 56, 56, 256), (40, 14, 14, 1280), (5, 7, 7, 2048),
                                     (2, 28, 28, 512), (1, 3, 5, 256)])
def test_dwconv3x3_bf16_vs_torch(B, H, W, C):
    """csrc/dwconv.hip (PVTv2 DWConv) forward / backward-data / weight + bias gradients against F.conv2d in fp32 on the
    same bf16-rounded inputs."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.dwconv import dwconv3x3
    torch.manual_seed(B * C + H)
    x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(C, 1, 3, 3, device="cuda") * 0.3).requires_grad_(True)
    b = torch.randn(C, device="cuda").requires_grad_(True)
    y = dwconv3x3(x, w, b)
    g = torch.randn_like(y)
    dx, dw, db = torch.autograd.grad(y, (x, w, b), g)
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, br, 1, 1, groups=C)
    dxr, dwr, dbr = torch.autograd.grad(yr, (xr, wr, br), g.float().permute(0, 3, 1, 2))
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and dw.dtype == torch.float32
    assert (y.float() - yr.permute(0, 2, 3, 1)).abs().max() <= 2e-2 * yr.abs().max()          # bf16 output rounding
    assert (dx.float() - dxr.permute(0, 2, 3, 1)).abs().max() <= 2e-2 * dxr.abs().max()
    assert (dw - dwr).abs().max() <= 2e-4 * dwr.abs().max() + 1e-3                               # fp32 accumulation
    assert (db - dbr).abs().max() <= 2e-4 * dbr.abs().max() + 1e-3


def test_pvt_batched_casts_match_autocast():
    """PVTv2 with all parameter casts done by one autograd node + explicit dtype rules against plain per-module autocast."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.backbone_pvt import PyramidVisionTransformerV2
    torch.manual_seed(0)
    m = PyramidVisionTransformerV2(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), qkv_bias=True, norm_eps=1e-6,
                                   depths=(2, 2, 3, 2), drop_path_rate=0.0).cuda().train()
    x = torch.randn(2, 3, 96, 96, device="cuda")
    params = [p for p in m.parameters()]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = m(x)                 # batched casts, explicit precision
        b = m._forward(x, None)  # per-module autocast
    for k in a:
        assert a[k].dtype == b[k].dtype
        assert (a[k].float() - b[k].float()).abs().max() <= 0.03 * b[k].float().abs().max(), k
    ga = torch.autograd.grad(sum(v.float().pow(2).mean() for v in a.values()), params, allow_unused=True)
    gb = torch.autograd.grad(sum(v.float().pow(2).mean() for v in b.values()), params, allow_unused=True)
    r = m._forward(x, None)  # fp32 reference: tells which gradients are well determined at bf16 precision at all
    gr = torch.autograd.grad(sum(v.float().pow(2).mean() for v in r.values()), params, allow_unused=True)
    assert all(g.dtype == torch.float32 for g in ga if g is not None)

    def cos(u, v):
        return float(torch.nn.functional.cosine_similarity(u.flatten().float(), v.flatten().float(), dim=0))
    checked = 0
    for u, v, w in zip(ga, gb, gr):
        if u is None or w is None or float(w.abs().max()) == 0:
            continue
        if cos(v, w) > 0.9:  # (e.g. key biases have a zero true gradient: both bf16 paths return rounding noise there)
            assert cos(u, w) > 0.85, (cos(u, w), cos(v, w))
            checked += 1
    assert checked > len(params) // 4  # (at the 0.02-std initialisation about half of the gradients are below bf16 noise)
    fa = torch.cat([g.flatten() for g in ga if g is not None]); fb = torch.cat([g.flatten() for g in gb if g is not None])
    assert float((fa - fb).norm() / fb.norm()) < 0.2


@pytest.mark.parametrize("B,C,H,W,relu", [(3, 256, 14, 14, True), (2, 256, 56, 56, False), (4, 64, 7, 9, True)])
def test_groupnorm_nhwc_vs_torch(B, C, H, W, relu):
    """csrc/groupnorm.hip (GroupNorm(32) [+ ReLU] on channels_last maps) against nn.GroupNorm (+ relu) in fp64."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.groupnorm import group_norm
    torch.manual_seed(C + H)
    gn = torch.nn.GroupNorm(32, C).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5); gn.bias.normal_(0, 0.3)
    x = (torch.randn(B, C, H, W, device="cuda") * 2 + 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = group_norm(x, gn, relu=relu)
    assert y.is_contiguous(memory_format=torch.channels_last)
    g = torch.randn_like(y)
    dx, dw, db = torch.autograd.grad(y, (x, gn.weight, gn.bias), g)
    x64 = x.detach().double().contiguous().requires_grad_(True)
    w64, b64 = gn.weight.detach().double().requires_grad_(True), gn.bias.detach().double().requires_grad_(True)
    y64 = torch.nn.functional.group_norm(x64, 32, w64, b64, gn.eps)
    if relu:
        y64 = y64.relu()
    dx64, dw64, db64 = torch.autograd.grad(y64, (x64, w64, b64), g.double())
    assert (y.double() - y64).abs().max() < 2e-5
    assert (dx.double() - dx64).abs().max() <= 2e-5 * dx64.abs().max() + 1e-6
    assert (dw.double() - dw64).abs().max() <= 2e-5 * dw64.abs().max() + 1e-5
    assert (db.double() - db64).abs().max() <= 2e-5 * db64.abs().max() + 1e-5


@pytest.mark.parametrize("B,C,H,W", [(3, 256, 28, 28), (2, 8, 5, 7), (1, 64, 1, 3)])
def test_upsample2x_bilinear_nhwc_vs_torch(B, C, H, W):
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.upsample import upsample_bilinear
    torch.manual_seed(H * W)
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = upsample_bilinear(x, (2 * H, 2 * W))
    xr = x.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.interpolate(xr, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    assert (y - yr).abs().max() < 1e-6
    g = torch.randn_like(yr)
    dx, = torch.autograd.grad(y, x, g.contiguous(memory_format=torch.channels_last))
    dxr, = torch.autograd.grad(yr, xr, g)
    assert (dx - dxr).abs().max() <= 1e-5 * dxr.abs().max() + 1e-6


@pytest.mark.parametrize("B,C,H,W", [(3, 256, 28, 28), (2, 8, 5, 7)])
def test_fpn_step_lateral_plus_upsampled_in_one_pass(B, C, H, W):
    """ops.upsample.upsample_bilinear_add(lateral, x) == lateral + F.interpolate(x) (msdeformattn.py:350): bitwise the two-pass
    result of the own kernels, within round-off of ATen's; the lateral branch receives dy itself, x the gathered gradient."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.upsample import upsample_bilinear, upsample_bilinear_add
    torch.manual_seed(H + W)
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lat = torch.randn(B, C, 2 * H, 2 * W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = upsample_bilinear_add(lat, x)
    two = lat + upsample_bilinear(x, (2 * H, 2 * W))
    assert torch.equal(y, two)
    ref = lat + torch.nn.functional.interpolate(x, size=(2 * H, 2 * W), mode="bilinear", align_corners=False)
    assert (y - ref).abs().max() < 1e-6
    g = torch.randn_like(ref).contiguous(memory_format=torch.channels_last)
    dlat, dx = torch.autograd.grad(y, (lat, x), g)
    dlat_r, dx_r = torch.autograd.grad(ref, (lat, x), g)
    assert torch.equal(dlat, dlat_r)
    assert (dx - dx_r).abs().max() <= 1e-5 * dx_r.abs().max() + 1e-6
    # a lateral map that is not channels_last takes the two-pass route and gives the same values
    lat2 = lat.detach().contiguous()
    assert (upsample_bilinear_add(lat2, x.detach()) - ref).abs().max() < 1e-6


@pytest.mark.parametrize("rows,C,with_res", [(41160, 256, True), (4000, 256, True), (4000, 256, False), (37, 128, True), (5, 512, False)])
def test_add_layernorm_forward_backward_vs_torch(rows, C, with_res):
    """ops.layernorm.LayerNorm(x, residual): LN(x + r) in one pass (csrc/layernorm.hip) == nn.LayerNorm(x + r) in float64,
    values, input gradients (the same tensor serves both branches) and the parameter gradients, immediate and deferred
    (reference: post-norm layers of msdeformattn.py:119-134 and transformer_decoder.py:50-58, 99-118, 178-182)."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    from combo_avs_amd.ops.layernorm import LayerNorm
    torch.manual_seed(rows + C)
    ln = LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(C))
        ln.bias.copy_(0.1 * torch.randn(C))
    x = (torch.randn(rows, C, device="cuda") * 2 + 0.5).requires_grad_(True)
    r = torch.randn(rows, C, device="cuda", requires_grad=True) if with_res else None
    g = torch.randn(rows, C, device="cuda")
    xd = x.detach().double().requires_grad_(True)
    rd = r.detach().double().requires_grad_(True) if with_res else None
    wd, bd = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd + rd if with_res else xd, (C,), wd, bd, ln.eps)
    ref_g = torch.autograd.grad(ref, [xd] + ([rd] if with_res else []) + [wd, bd], g.double())
    for deferred in (False, True):
        ln.defer_dw = deferred
        inputs = [x] + ([r] if with_res else []) + [ln.weight, ln.bias]
        if deferred:
            with L.deferred_dw():
                y = ln(x, r)
                got = torch.autograd.grad(y, inputs, g)
        else:
            y = ln(x, r)
            got = torch.autograd.grad(y, inputs, g)
        assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 1e-6
        for a, b in zip(got, ref_g):
            assert ((a.double() - b).norm() / b.norm()).item() < 2e-6


def test_backbone_fp32_weight_gradients_on_the_grouped_kernels():
    """fp32 ResNet: the weight gradients of the stride-1 convolutions as problems of the deferred grouped launch / the
    implicit-GEMM 3x3 kernel (ops/convwrw.py, FrozenBN scale applied after the flush) against the library's backward."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.backbone import ResNet
    from combo_avs_amd.ops import convwrw
    from combo_avs_amd.ops import linear as L
    torch.manual_seed(0)
    m = ResNet(50).cuda().train()
    for mod in m.modules():
        if hasattr(mod, "running_var"):
            mod.running_var.uniform_(0.5, 1.5); mod.running_mean.normal_(0, 0.1); mod.weight.uniform_(0.8, 1.2); mod.bias.normal_(0, 0.1)
    m._bn_cache.clear()
    x = torch.randn(6, 3, 128, 128, device="cuda")
    params = [p for p in m.parameters()]

    def grads(enabled):
        convwrw.ENABLED = enabled
        try:
            with L.deferred_dw():
                out = m(x)
                g = torch.autograd.grad(sum(v.pow(2).mean() for v in out.values()), params)
            return [t.clone() for t in g]
        finally:
            convwrw.ENABLED = True
    # judge both against float64 (the library itself moves by ~1e-3 on the stem weight between algorithms it picks)
    import copy
    m64 = copy.deepcopy(m).double()

    def unfused(mm, xx):  # per-convolution torch path (conv + FrozenBN affine + relu), no folded weights
        xx = torch.nn.functional.max_pool2d(torch.relu(mm.stem.conv1(xx)), 3, 2, 1)
        res = {}
        for name in ("res2", "res3", "res4", "res5"):
            for blk in getattr(mm, name):
                xx = blk(xx)
            res[name] = xx
        return res
    out64 = unfused(m64, x.double())
    true = torch.autograd.grad(sum(v.pow(2).mean() for v in out64.values()), [p for p in m64.parameters()])
    # (measured: two runs of the LIBRARY path differ from float64 by 1.2e-3 .. 2.9e-3 on the stem weight - atomics order in
    # its weight-gradient kernels + ReLU gates next to 0 - so per-parameter bounds are loose and the whole-vector error decides)
    tn = torch.sqrt(sum(b.pow(2).sum() for b in true))

    def err(gs):
        worst = max(float((a.double() - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(gs, true))
        return worst, float(torch.sqrt(sum((a.double() - b).pow(2).sum() for a, b in zip(gs, true))) / tn)
    # (the library's FIRST call of a configuration returns the result of another, more accurate algorithm than the one its
    # find pass settles on for later calls: compare steady states)
    grads(False), grads(True)
    (w_lib, e_lib), (w_own, e_own) = err(grads(False)), err(grads(True))
    assert len(L._after_flush) == 0
    # plumbing check (a missing / doubled FrozenBN scale, an unwritten deferred gradient would be O(0.1 .. 1)); the kernels'
    # own accuracy is pinned per layer by test_backbone_conv_weight_gradient_kernels_vs_fp64
    assert e_own < max(3 * e_lib, 2e-3) and w_own < max(3 * w_lib, 3e-2), (e_own, e_lib, w_own, w_lib)
    # without a deferred_dw() context the same path computes every gradient at once
    convwrw.ENABLED = True
    out = m(x)
    g2 = torch.autograd.grad(sum(v.pow(2).mean() for v in out.values()), params)
    w2, e2 = err(g2)
    assert e2 < max(3 * e_lib, 2e-3) and w2 < max(3 * w_lib, 3e-2), (e2, e_lib, w2, w_lib)


@pytest.mark.parametrize("B,C,Co,H,k", [(6, 256, 64, 32, 1), (6, 64, 256, 32, 1), (6, 1024, 256, 8, 1), (6, 512, 2048, 4, 1),
                                        (6, 128, 128, 16, 3), (6, 256, 256, 8, 3), (6, 512, 512, 4, 3)])
def test_backbone_conv_weight_gradient_kernels_vs_fp64(B, C, Co, H, k):
    """one convolution through ops.convwrw.conv2d: dW on the head's kernels (immediately, and as a problem of the deferred
    grouped launch), dX by the library (3x3) / the 3-product GEMM (1x1) - both against float64"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import convwrw
    from combo_avs_amd.ops import linear as L
    torch.manual_seed(B * C + Co + H + k)
    x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Co, C, k, k, device="cuda") * 0.05).requires_grad_(True)
    g = torch.randn(B, Co, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    assert convwrw.kind(x, w, 1, k // 2) == k
    rx, rw = torch.autograd.grad(torch.nn.functional.conv2d(x.double(), w.double(), None, 1, k // 2), (x, w), g.double())
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    dx, dw = torch.autograd.grad(convwrw.conv2d(x, w, 1, k // 2), (x, w), g)
    # (dX of the 1x1 layers also runs on the 3-product kernels since ops.convwrw.DX_OWN = 2; the 3x3 dX is the library's)
    assert rel(dx, rx) < 2e-5 and rel(dw, rw) < 2e-5, (rel(dx, rx), rel(dw, rw))
    with L.deferred_dw():
        dx2, dw2 = torch.autograd.grad(convwrw.conv2d(x, w, 1, k // 2), (x, w), g)
    assert rel(dx2, rx) < 2e-5 and rel(dw2, rw) < 2e-5, (rel(dx2, rx), rel(dw2, rw))


def test_layernorm_fanout_aliases_and_pos_output_match_torch():
    """ops/layernorm.py fan-out: LN(x + r) handed out as aliases (one autograd output per consumer) plus `+ pos`; the consumers'
    gradients are summed inside the backward kernel - against nn.LayerNorm with autograd's own accumulation."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.layernorm import LayerNorm
    torch.manual_seed(0)
    B, S, C = 3, 37, 256
    ln = LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_()
    ref = torch.nn.LayerNorm(C).cuda()
    ref.load_state_dict(ln.state_dict())
    x = torch.randn(B, S, C, device="cuda", requires_grad=True)
    r = torch.randn(B, S, C, device="cuda", requires_grad=True)
    # pos is LEARNABLE at every call site (query_embed.weight; sine rows + level_embed): its gradient must come back
    pos = torch.randn(1, S, C, device="cuda", requires_grad=True)
    w1, w2, w3 = (torch.randn(B, S, C, device="cuda") for _ in range(3))
    a, b, q = ln(x, r, fanout=2, pos=pos)
    assert a.data_ptr() == b.data_ptr() and q.data_ptr() != a.data_ptr()
    loss = (a * w1).sum() + (b * w2).sum() + (q * w3).sum()
    got = torch.autograd.grad(loss, [x, r, ln.weight, ln.bias, pos])
    x2, r2 = x.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
    pos2 = pos.detach().clone().requires_grad_(True)
    y = ref(x2 + r2)
    loss2 = (y * w1).sum() + (y * w2).sum() + ((y + pos2) * w3).sum()
    want = torch.autograd.grad(loss2, [x2, r2, ref.weight, ref.bias, pos2])
    assert got[4].shape == pos.shape
    torch.testing.assert_close(a, y, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(q, y + pos, rtol=1e-5, atol=1e-5)
    for g, wnt in zip(got, want):
        torch.testing.assert_close(g, wnt, rtol=2e-4, atol=2e-4)
    # one consumer unused: its gradient arrives as None
    a, b, q = ln(x, r, fanout=2, pos=pos)
    got = torch.autograd.grad((a * w1).sum() + (q * w3).sum(), [x])
    y = ref(x2 + r2)
    want = torch.autograd.grad((y * w1).sum() + ((y + pos) * w3).sum(), [x2])
    torch.testing.assert_close(got[0], want[0], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("HW,hw,BT,Q", [((56, 56), (7, 7), 5, 100), ((56, 56), (14, 14), 5, 100), ((56, 56), (28, 28), 3, 100),
                                         ((128, 128), (64, 64), 2, 100), ((40, 56), (9, 13), 2, 37)])
def test_fused_mask_bits_equal_the_reference_rule(HW, hw, BT, Q):
    """csrc/maskbits.hip: attention mask from mask_embed and the DOWNSAMPLED pixel embedding (one fp32-MFMA kernel, ballots ->
    bit words, row reset) against the reference's formulation - full einsum, F.interpolate, sigmoid < 0.5, row reset
    (transformer_decoder.py:458, 498-507) - in fp64 on the same operands: cells may differ only where the interpolated logit is
    within 1e-5 x RMS of 0; and against the same rule applied by torch to the product's own full-resolution logits."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import masklogit
    torch.manual_seed(HW[0] + hw[0])
    C = 256
    me = torch.randn(BT, Q, C, device="cuda")
    me[0, 3] = -me[0, 3].abs() * 0 - 1.0  # a query whose logits are all negative against positive features: fully blocked row
    mf = torch.randn(BT, HW[0] * HW[1], C, device="cuda")
    mf[0] = mf[0].abs()
    mfd = masklogit.downsample_tokens(mf, HW, hw)
    ref_d = torch.nn.functional.interpolate(mf.transpose(1, 2).reshape(BT, C, *HW), size=hw, mode="bilinear", align_corners=False)
    torch.testing.assert_close(mfd, ref_d.flatten(2).transpose(1, 2), rtol=1e-6, atol=1e-6)
    pm = masklogit.mask_bits(me, mfd, True, with_bytes=True)
    n = hw[0] * hw[1]
    got = pm.bytes[:, :, :n].bool()
    logits = torch.einsum("bqc,bpc->bqp", me.double(), mf.double()).view(BT, Q, *HW)
    down = torch.nn.functional.interpolate(logits, size=hw, mode="bilinear", align_corners=False).flatten(2)
    want = down.float().sigmoid() < 0.5
    full = want.all(-1)
    assert bool(full[0, 3]) and int(full.sum()) >= 1
    want[full] = False
    near = down.abs() < 1e-5 * down.pow(2).mean().sqrt()
    near = near | near.any(-1, keepdim=True) & full[..., None]  # (a row whose "fully blocked" verdict hinges on a near-zero cell)
    assert bool(((got == want) | near).all()), int(((got != want) & ~near).sum())
    assert bool((pm.bytes[:, :, n:] == 1).all())
    # bit rows == byte rows, padding bits blocked
    inj = masklogit.pack_mask(got, reset_full_rows=False)
    assert torch.equal(inj.bits, pm.bits)
    # the two-step formulation on the product's own full-resolution fp32 logits (csrc/gemm_f32.hip) agrees away from 0
    out = torch.empty(BT, Q, HW[0] * HW[1], device="cuda")
    masklogit.mask_logits_into(me, mf, out)
    own = torch.nn.functional.interpolate(out.view(BT, Q, *HW), size=hw, mode="bilinear", align_corners=False).flatten(2)
    old = own.sigmoid() < 0.5
    old[old.all(-1)] = False
    assert bool(((old == got) | near).all())


def test_pack_mask_equals_the_mask_kernel():
    """ops.masklogit.pack_mask (the helper tests use to inject masks) writes what csrc/maskbits.hip writes for the same
    decisions: byte rows padded with blocked cells, bit k of word j = key 32 j + k, fully blocked rows un-blocked"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import masklogit
    torch.manual_seed(3)
    BT, Q, C = 3, 100, 256
    for hw in (49, 196, 784, 117):
        me = torch.randn(BT, Q, C, device="cuda")
        mfd = torch.randn(BT, hw, C, device="cuda")
        me[0, 5] = 0.0
        me[0, 5, 0] = -50.0
        mfd[0, :, 0] = mfd[0, :, 0].abs() + 0.1  # query 5 of frame 0: every cell blocked -> reset to all-open (:458)
        ker = masklogit.mask_bits(me, mfd, True, with_bytes=True)
        raw = masklogit.mask_bits(me, mfd, False, with_bytes=True)
        assert raw.bytes[0, 5, :hw].all() and not ker.bytes[0, 5, :hw].any()
        inj = masklogit.pack_mask(raw.bytes[:, :, :hw].view(torch.bool), True)
        assert torch.equal(inj.bytes, ker.bytes) and torch.equal(inj.bits, ker.bits)


def test_mask_logits_of_all_heads_in_one_launch_equal_per_head_launches():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import masklogit
    torch.manual_seed(1)
    heads, BT, Q, HW, C = 10, 5, 100, 3136, 256
    mes = [torch.randn(BT, Q, C, device="cuda") for _ in range(heads)]
    mf = torch.randn(BT, HW, C, device="cuda")
    from combo_avs_amd.ops import linear as L
    allb = torch.empty(heads, BT, Q, HW, device="cuda")
    L.set_forward_precision("fp32")  # the exact path (`--head-dtype fp32`): ONE launch of csrc/gemm_f32.hip for all heads
    try:
        masklogit.mask_logits_all_into(mes, mf, allb)
    finally:
        L.set_forward_precision(L.DEFAULT_FORWARD_PRECISION)
    one = torch.empty(BT, Q, HW, device="cuda")
    for h in range(heads):
        masklogit.mask_logits_into(mes[h], mf, one)
        assert torch.equal(one, allb[h])  # the same kernel, the same k order: bit for bit
    # the default forward mode (3 fp16-piece products on csrc/gemm_nt3.hip, per head): the same values to fp32 round-off
    dflt = torch.empty(heads, BT, Q, HW, device="cuda")
    masklogit.mask_logits_all_into(mes, mf, dflt)
    assert float((dflt - allb).norm() / allb.norm()) < 1e-6
    ref = torch.stack([m.double() @ mf.double().transpose(1, 2) for m in mes])
    assert float((dflt.double() - ref).norm() / ref.norm()) <= 1.5 * float((allb.double() - ref).norm() / ref.norm())


# ---- channel sums (csrc/colsum.hip): the bias / level-embedding gradients that must not go through ATen's split reduction ----
@pytest.mark.gpu
@pytest.mark.parametrize("shape,cd,dt", [
    ((7840, 320), -1, torch.bfloat16), ((7840, 1280), -1, torch.bfloat16), ((125440, 64), -1, torch.bfloat16),
    ((31360, 512), -1, torch.bfloat16), ((3000, 256), -1, torch.float32), ((37, 12), -1, torch.float32),
    ((40, 320, 14, 14), 1, torch.bfloat16), ((40, 64, 56, 56), 1, torch.bfloat16), ((40, 256, 784), 1, torch.float32),
    ((40, 256, 49), 1, torch.float32), ((3, 5, 7), 1, torch.float32), ((1, 256, 56, 56), 1, torch.float32)])
def test_channel_sum_matches_float64_and_is_deterministic(shape, cd, dt):
    from combo_avs_amd.ops import colsum
    torch.manual_seed(len(shape) * 131 + shape[0])
    x = torch.randn(shape, device="cuda").to(dt)
    got = colsum.sum_to_channels(x, cd, out_dtype=torch.float32)
    dims = [d for d in range(x.dim()) if d != cd % x.dim()]
    want = x.double().sum(dims)
    n = x.numel() // want.numel()
    # fp32 accumulation of n terms of unit variance: error ~ 1e-7 * sqrt(n) * a few (the stated tolerance)
    assert float((got.double() - want).abs().max()) <= 4e-6 * n ** 0.5 + 1e-6
    again = colsum.sum_to_channels(x, cd, out_dtype=torch.float32)
    assert torch.equal(got, again)  # fixed summation order


@pytest.mark.gpu
def test_channel_sum_channels_last_and_autograd_wrappers():
    from combo_avs_amd.ops import colsum
    torch.manual_seed(5)
    x = torch.randn(6, 64, 9, 11, device="cuda").to(memory_format=torch.channels_last)
    got = colsum.sum_to_channels(x, 1)
    assert torch.allclose(got, x.sum((0, 2, 3)), atol=1e-4)
    # y = x + v[channel]: dv against autograd's own reduction
    for shape, cd in (((4, 32, 10, 10), 1), ((4, 32, 50), 1)):
        xx = torch.randn(shape, device="cuda", requires_grad=True)
        v = torch.randn(32, device="cuda", requires_grad=True)
        g = torch.randn(shape, device="cuda")
        colsum.add_channel_vector(xx, v, cd).backward(g)
        view = [1, -1] + [1] * (len(shape) - 2)
        x2, v2 = xx.detach().clone().requires_grad_(), v.detach().clone().requires_grad_()
        (x2 + v2.view(view)).backward(g)
        assert torch.equal(xx.grad, x2.grad) and torch.allclose(v.grad, v2.grad, atol=1e-4)
    # linear with bias: all three gradients against F.linear's
    a = torch.randn(5, 77, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    w = torch.randn(128, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    b = torch.randn(128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    gy = torch.randn(5, 77, 128, device="cuda", dtype=torch.bfloat16)
    y = colsum.linear_bias(a, w, b)
    y.backward(gy)
    a2, w2, b2 = (t.detach().clone().requires_grad_() for t in (a, w, b))
    y2 = torch.nn.functional.linear(a2, w2, b2)
    y2.backward(gy)
    assert torch.equal(y, y2)
    for p, q in ((a, a2), (w, w2), (b, b2)):  # bf16 results of fp32-accumulated sums in different orders: 1 bf16 ulp
        assert float((p.grad.float() - q.grad.float()).abs().max()) <= 2 ** -7 * float(q.grad.float().abs().max())


@pytest.mark.gpu
def test_channel_sum_survives_hipgraph_replays_where_the_library_reduction_does_not():
    """the hazard itself is documented by tools/graph_reduce_repro.py; this pins the replacement: 30 replays with an eager kernel
    between two synchronisations before each, every result equal to the eager one"""
    from combo_avs_amd.ops import colsum
    torch.manual_seed(1)
    xs = [torch.randn(15680, 1280, device="cuda").bfloat16(), torch.randn(8, 5376, 256, device="cuda"),
          torch.randn(40, 256, 28, 28, device="cuda")]
    cds = [-1, -1, 1]
    want = [colsum.sum_to_channels(x, cd).clone() for x, cd in zip(xs, cds)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        [colsum.sum_to_channels(x, cd) for x, cd in zip(xs, cds)]
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = [[colsum.sum_to_channels(x, cd) for x, cd in zip(xs, cds)] for _ in range(4)]
    junk = torch.ones(64, device="cuda")
    for _ in range(30):
        torch.cuda.synchronize()
        junk[:16].zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for row in outs:
            for o, w in zip(row, want):
                assert torch.equal(o, w)


@pytest.mark.gpu
def test_row_sum_forward_backward():
    from combo_avs_amd.ops import colsum
    torch.manual_seed(2)
    x = torch.randn(10, 4000, device="cuda", requires_grad=True)
    g = torch.randn(10, device="cuda")
    y = colsum.row_sum(x)
    y.backward(g)
    assert torch.allclose(y, x.detach().double().sum(1).float(), atol=1e-4)
    assert torch.equal(x.grad, g[:, None].expand(-1, 4000))


@pytest.mark.gpu
def test_channel_sum_timing(capsys):
    """HBM-bound: the input is read once.  Printed, not asserted (host-timed on a shared box)."""
    from combo_avs_amd.ops import colsum
    for shape, cd, dt in (((125440, 256), -1, torch.float32), ((125440, 64), -1, torch.bfloat16), ((7840, 1280), -1, torch.bfloat16),
                          ((31360, 512), -1, torch.bfloat16), ((40, 256, 28, 28), 1, torch.float32), ((520, 512), -1, torch.float32)):
        x = torch.randn(shape, device="cuda").to(dt)
        dims = [d for d in range(x.dim()) if d != cd % x.dim()]

        def t(fn, n=30):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / n * 1e3
        own = min(t(lambda: colsum.sum_to_channels(x, cd)) for _ in range(3))
        lib = min(t(lambda: x.sum(dims)) for _ in range(3))
        gb = x.numel() * x.element_size() / 1e9
        with capsys.disabled():
            print(f"\n[colsum {tuple(shape)} {str(dt)[6:]}] {own:.1f} us = {gb / own * 1e6 / 1e3:.2f} TB/s (library sum: {lib:.1f} us)")


# ---- pre-norm residual step of the PVTv2 blocks (csrc/prenorm.hip) ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("C,N,with_r,with_scale,out_fp32", [(64, 50, True, True, False), (128, 33, True, False, False),
                                                            (320, 21, False, False, False), (512, 7, True, True, True),
                                                            (256, 19, True, True, False)])
def test_prenorm_step_matches_float64(C, N, with_r, with_scale, out_fp32):
    from combo_avs_amd.ops.prenorm import _PreNorm
    torch.manual_seed(C + N)
    B = 5
    x = torch.randn(B, N, C, device="cuda", requires_grad=True)
    r = (torch.randn(B, N, C, device="cuda") * 0.5).bfloat16().requires_grad_() if with_r else None
    scale = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25], device="cuda") if with_scale else None
    w = (torch.randn(C, device="cuda") * 0.3 + 1).requires_grad_()
    b = (torch.randn(C, device="cuda") * 0.1).requires_grad_()
    out = _PreNorm.apply(x, r, scale, w, b, 1e-6, out_fp32, False)
    z, y = out if with_r else (None, out)
    assert y.dtype == (torch.float32 if out_fp32 else torch.bfloat16)
    # float64 reference
    xd, wd, bd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    rd = r.detach().double().requires_grad_() if with_r else None
    zd = xd if not with_r else xd + (scale.double()[:, None, None] if with_scale else 1.0) * rd
    yd = torch.nn.functional.layer_norm(zd, (C,), wd, bd, 1e-6)
    if with_r:
        assert torch.allclose(z.double(), zd, atol=1e-6)
    tol = 2e-6 if out_fp32 else 2 ** -8
    assert float((y.double() - yd).abs().max()) <= tol * float(yd.abs().max()) + 1e-6
    gy = torch.randn(B, N, C, device="cuda").to(y.dtype)
    gz = torch.randn(B, N, C, device="cuda") if with_r else None
    torch.autograd.backward([t for t in (z, y) if t is not None], [g for g in (gz, gy) if g is not None])
    loss = (yd * gy.double()).sum() + ((zd * gz.double()).sum() if with_r else 0.0)
    loss.backward()
    assert float((x.grad.double() - xd.grad).abs().max()) <= 1e-5 * float(xd.grad.abs().max()) + 1e-6
    if with_r:
        assert r.grad.dtype == torch.bfloat16
        assert float((r.grad.double() - rd.grad).abs().max()) <= 2 ** -8 * float(rd.grad.abs().max()) + 1e-6
        if with_scale:
            assert float(r.grad[0].abs().max()) == 0.0  # a dropped sample passes no gradient to its branch
    assert float((w.grad.double() - wd.grad).abs().max()) <= 1e-4 * float(wd.grad.abs().max()) + 1e-5
    assert float((b.grad.double() - bd.grad).abs().max()) <= 1e-4 * float(bd.grad.abs().max()) + 1e-5


@pytest.mark.gpu
def test_prenorm_fanout_sums_both_consumers_gradients():
    from combo_avs_amd.ops.prenorm import _PreNorm
    torch.manual_seed(3)
    B, N, C = 3, 17, 128
    x = torch.randn(B, N, C, device="cuda", requires_grad=True)
    r = torch.randn(B, N, C, device="cuda").bfloat16().requires_grad_()
    w = torch.ones(C, device="cuda", requires_grad=True)
    b = torch.zeros(C, device="cuda", requires_grad=True)
    z, y1, y2 = _PreNorm.apply(x, r, None, w, b, 1e-6, False, False, 2)
    assert y1.data_ptr() == y2.data_ptr()
    g1, g2 = torch.randn_like(y1), torch.randn_like(y1)
    torch.autograd.backward([y1, y2], [g1, g2])
    got = [t.grad.clone() for t in (x, r, w, b)]
    for t in (x, r, w, b):
        t.grad = None
    z, y = _PreNorm.apply(x, r, None, w, b, 1e-6, True, False, 1)  # fp32 output: takes the exact fp32 sum of the two bf16
    y.backward(g1.float() + g2.float())                              # gradients, which is what the fan-out kernel forms
    for a, t in zip(got, (x, r, w, b)):
        assert torch.allclose(a.float(), t.grad.float(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_pvt_prenorm_path_matches_per_op_path():
    """PVTv2 (bf16 training path) with the fused pre-norm residual steps against the per-op formulation, stochastic depth off:
    same outputs and gradients up to bf16 rounding; with stochastic depth on the multipliers are 0 or 1 / keep per sample."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import backbone_pvt as BP
    torch.manual_seed(0)
    m = BP.PyramidVisionTransformerV2(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), qkv_bias=True, norm_eps=1e-6,
                                      depths=(2, 2, 3, 2), drop_path_rate=0.0).cuda().train()
    x = torch.randn(2, 3, 96, 96, device="cuda")
    params = [p for p in m.parameters()]

    def run(flag):
        old, BP.PRENORM = BP.PRENORM, flag
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = m(x)
            g = torch.autograd.grad(sum(v.float().pow(2).mean() for v in out.values()), params, allow_unused=True)
        finally:
            BP.PRENORM = old
        return out, g
    a, ga = run(True)
    b, gb = run(False)
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype
        assert float((a[k].float() - b[k].float()).abs().max()) <= 0.03 * float(b[k].float().abs().max()), k
    fa = torch.cat([g.flatten() for g in ga if g is not None])
    fb = torch.cat([g.flatten() for g in gb if g is not None])
    assert fa.numel() == fb.numel() and bool(torch.isfinite(fa).all())
    assert float((fa - fb).norm() / fb.norm()) < 0.1
    # stochastic depth: the table of multipliers
    m2 = BP.PyramidVisionTransformerV2(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), qkv_bias=True, norm_eps=1e-6,
                                       depths=(2, 2, 3, 2), drop_path_rate=0.3).cuda().train()
    sc = m2._drop_path_scales(64, torch.device("cuda"))
    assert sc.shape == (18, 64)
    keep = 1.0 - torch.tensor([0.3 * i / 8 for i in range(9)]).repeat_interleave(2)[:, None].cuda()
    assert bool(((sc == 0) | ((sc - 1.0 / keep).abs() < 1e-6)).all())
    assert float(sc[0].min()) == 1.0 and 0.5 < float((sc[-1] > 0).float().mean()) < 0.9  # first block never dropped, last ~30 %
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m2(x)
    torch.autograd.grad(sum(v.float().pow(2).mean() for v in out.values()), [p for p in m2.parameters()], allow_unused=True)
    m2.eval()
    assert m2._drop_path_scales(4, torch.device("cuda")) is None


@pytest.mark.gpu
@pytest.mark.parametrize("C,rows", [(64, 49 * 3), (320, 49 * 5), (128, 11)])
def test_bias_ln_bf16_matches_float64(C, rows):
    from combo_avs_amd.ops.prenorm import _BiasLn
    torch.manual_seed(C)
    x = torch.randn(1, rows, C, device="cuda").bfloat16().requires_grad_()
    xb = (torch.randn(C, device="cuda") * 0.5).bfloat16().requires_grad_()
    w = (torch.randn(C, device="cuda") * 0.3 + 1).requires_grad_()
    b = (torch.randn(C, device="cuda") * 0.1).requires_grad_()
    y = _BiasLn.apply(x, xb, w, b, 1e-5, False)
    g = torch.randn_like(y)
    y.backward(g)
    xd, xbd, wd, bd = (t.detach().double().requires_grad_() for t in (x, xb, w, b))
    yd = torch.nn.functional.layer_norm(xd + xbd, (C,), wd, bd, 1e-5)
    yd.backward(g.double())
    assert y.dtype == torch.bfloat16 and x.grad.dtype == torch.bfloat16 and xb.grad.dtype == torch.bfloat16
    assert float((y.double() - yd).abs().max()) <= 2 ** -8 * float(yd.abs().max()) + 1e-6
    assert float((x.grad.double() - xd.grad).abs().max()) <= 2 ** -8 * float(xd.grad.abs().max()) + 1e-6
    # the bias gradient is the channel sum of the bf16-rounded dx: error ~ sqrt(rows) * 2^-9 * max|dx|
    assert float((xb.grad.double() - xbd.grad).abs().max()) <= 2 ** -7 * rows ** 0.5 * float(xd.grad.abs().max()) + 1e-3
    assert float((w.grad.double() - wd.grad).abs().max()) <= 1e-4 * float(wd.grad.abs().max()) + 1e-5
    assert float((b.grad.double() - bd.grad).abs().max()) <= 1e-4 * float(bd.grad.abs().max()) + 1e-5


@pytest.mark.gpu
def test_deferred_grouped_column_sums_match_immediate_ones():
    from combo_avs_amd.ops import colsum
    torch.manual_seed(9)
    shapes = [(7840, 320), (7840, 1280), (31360, 128), (1960, 512), (49 * 40, 640), (125440, 64), (37, 12), (300, 2048)] * 7  # > 40 problems
    xs = [torch.randn(s, device="cuda").to(torch.bfloat16 if i % 3 else torch.float32) for i, s in enumerate(shapes)]
    q = colsum.DeferredColumnSums()
    outs = [q.add(x, x.dtype) for x in xs]
    assert len(q) > 0
    q.flush()
    assert not len(q)
    for x, o in zip(xs, outs):
        want = x.double().sum(0)
        tol = (2 ** -8 if x.dtype == torch.bfloat16 else 1e-6) * float(want.abs().max()) + 4e-6 * x.shape[0] ** 0.5
        assert float((o.double() - want).abs().max()) <= tol
    # the byte cap flushes by itself
    old, colsum.PENDING_CAP = colsum.PENDING_CAP, 1 << 20
    try:
        o = q.add(xs[1], torch.float32)
        assert not len(q) and torch.allclose(o, xs[1].float().sum(0), rtol=1e-3, atol=1e-2)
    finally:
        colsum.PENDING_CAP = old
    # a queue belongs to one backbone application = one autograd stream: feeding or flushing it from another stream raises
    q.add(xs[0], torch.float32)
    with torch.cuda.stream(torch.cuda.Stream()):
        with pytest.raises(RuntimeError, match="two HIP streams"):
            q.add(xs[1], torch.float32)
        with pytest.raises(RuntimeError, match="another HIP stream"):
            q.flush()
