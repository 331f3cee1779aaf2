"""GPU parity of the product head (pixel decoder -> AVFuse -> audio MLP -> masked decoder) and criterion against
the golden vectors generated from the reference (tests/golden/head.npz, criterion.npz, inference.npz).
SURVEY §8 rows a2-a5, a7-a17.  Weights/inputs are regenerated from synth.py; the reference's state-dict key list
stored in the fixture must match the product's keys exactly (checkpoint surface)."""
import json
import os

import numpy as np
import pytest
import torch

import gen_inputs
import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def build_head(num_classes=2):
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import combo_cfg
    from combo_avs_amd.backbone import ResNet
    from combo_avs_amd.meta_arch import build_sem_seg_head
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = combo_cfg(os.path.join(root, "configs/avs_s4/COMBO_R50_bs8_90k.yaml"))
    head = build_sem_seg_head(cfg, ResNet(50).output_shape())
    return head, cfg


@pytest.fixture(scope="module")
def head_run():
    z = np.load(os.path.join(G, "head.npz"))
    spec = json.loads(str(z["spec"]))
    head, cfg = build_head()
    ours = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    assert ours == {k: tuple(s) for k, s in spec}, "state-dict surface differs from the reference's"
    head.load_state_dict(synth.synth_state_dict(spec, 0))
    head = head.cuda().eval()
    feats, audio = gen_inputs.head_inputs()
    feats = {k: v.cuda().requires_grad_(True) for k, v in feats.items()}
    audio = audio.cuda().requires_grad_(True)
    out = head(dict(feats), audio)
    torch.cuda.synchronize()
    return z, head, feats, audio, out


def test_decoder_outputs_match_reference(head_run):
    """BASELINE.json north_star: mask logits within 1e-3 rel (fp32) of the reference - checked at the stated bound on the
    PRODUCT DEFAULT path, no outlier budget: every sampled mask logit of all 10 prediction heads within
    1e-3 * RMS(head) + 1e-3 * |ref| (the logits have RMS ~5; near-zero entries are judged against the head's scale).
    Forward dense layers run at fp32 grade (since round 6: three fp16-piece MFMA products per fp32 product, csrc/gemm_nt3.hip F16 - against
    float64 no more error than the exact instruction of csrc/gemm_f32.hip, which `COMBO_HEAD_FORWARD=fp32` selects; this test passes with
    either), so attention-mask cells flip only where fp32 round-off
    itself straddles 0 (transformer_decoder.py:493-509)."""
    z, head, feats, audio, out = head_run
    logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    got, ref = torch.stack(logits).detach().cpu().numpy(), z["dec/pred_logits"]
    bad = np.abs(got - ref) > 1e-3 * np.sqrt((ref ** 2).mean()) + 1e-3 * np.abs(ref)
    assert bad.sum() == 0, (bad.sum(), np.abs(got - ref).max())
    for i, m in enumerate(masks):
        d = synth.unpack(f"dec/pred_masks{i}", z)
        rms = float(d["l2"]) / np.sqrt(float(d["numel"]))
        synth.check_digest(m.cpu(), d, f"dec/pred_masks{i}", rtol=1e-3, atol=1e-3 * rms, frac_bad=0.0)
    assert len(out["middles_attn_mask"]) == 9 and out["middles_attn_mask"][0].shape == (5, 100, 3136)


def test_attention_masks_match_reference_bit_for_bit_up_to_round_off(head_run):
    """The thresholded masks themselves: the sign of the product's logits at the golden sample positions, and the reference rule
    `sigmoid(bilinear_down(logits)) < 0.5` applied by torch to the product's full-resolution logits against the masks the
    REFERENCE produced (head.npz `dec/attn_bits*`, as produced, before the row reset); cells may differ only where the
    (interpolated) logit lies within fp32 round-off of the threshold (counted)."""
    z, head, feats, audio, out = head_run
    ref_masks = synth.frozen_attn_masks(z)
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    flips = 0
    for i, m in enumerate(masks):
        d = synth.unpack(f"dec/pred_masks{i}", z)
        idx = synth.digest_indices(m.numel(), 4096, f"dec/pred_masks{i}")
        got = m.detach().reshape(-1).cpu().numpy()[idx].astype(np.float64)
        ref = np.asarray(d["sample"]).astype(np.float64)
        differ = (got < 0) != (ref < 0)
        rms = float(d["l2"]) / np.sqrt(float(d["numel"]))
        assert np.all(np.abs(ref[differ]) < 1e-5 * rms), (i, np.abs(ref[differ]).max())
        flips += int(differ.sum())
        if i < len(ref_masks):  # head i's mask gates layer i: the reference rule on the product's logits == the reference's mask
            tgt = [(7, 7), (14, 14), (28, 28)][i % 3]
            down = torch.nn.functional.interpolate(m.detach(), size=tgt, mode="bilinear", align_corners=False)
            blocked = (down.sigmoid().flatten(2) < 0.5).cpu()
            near = (down.flatten(2).abs() < 1e-5 * rms).cpu()
            assert bool(((blocked == ref_masks[i]) | near).all()), int(((blocked != ref_masks[i]) & ~near).sum())
    assert flips <= 2, flips


def test_intermediates_match_reference(head_run):
    z, head, feats, audio, out = head_run
    with torch.no_grad():
        mf, _, ms = head.pixel_decoder.forward_features({k: v.detach() for k, v in feats.items()})
        synth.check_digest(mf.cpu(), synth.unpack("pd/mask_features", z), "pd/mask_features", 1e-3, 2e-4)
        for i, m in enumerate(ms):
            synth.check_digest(m.cpu(), synth.unpack(f"pd/ms{i}", z), f"pd/ms{i}", 1e-3, 2e-4)
        fused = head.fusion_module({"res2": mf}, audio.detach())
        synth.check_digest(fused["visual"]["res2"].contiguous().cpu(), synth.unpack("fuse/visual", z), "fuse/visual", 1e-3, 2e-4)
        np.testing.assert_allclose(fused["audio"].cpu().numpy(), z["fuse/audio"], rtol=1e-3, atol=2e-4)
        a256 = head.audio_transformation(fused["audio"])
        np.testing.assert_allclose(a256.cpu().numpy(), z["fuse/audio256"], rtol=1e-3, atol=2e-4)


def test_inference_tail(head_run):
    from combo_avs_amd.meta_arch import MaskFormer
    z, head, feats, audio, out = head_run
    zi = np.load(os.path.join(G, "inference.npz"))
    with torch.no_grad():
        up = torch.nn.functional.interpolate(out["pred_masks"], size=(224, 224), mode="bilinear", align_corners=False)
        sem = torch.stack([MaskFormer.semantic_inference(None, c, m) for c, m in zip(out["pred_logits"], up)])
    synth.check_digest(sem.cpu(), synth.unpack("sem_seg", zi), "sem_seg", rtol=2e-3, atol=2e-3, frac_bad=0.005)


def make_criterion(mode):
    from combo_avs_amd.modeling.criterion import SetCriterion, SetCriterion_SS
    from combo_avs_amd.modeling.matcher import HungarianMatcher
    w = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 5.0, "loss_cosine": 10.0}
    wd = dict(w)
    for i in range(9):
        wd.update({f"{k}_{i}": v for k, v in w.items()})
    matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=12544)
    cls = SetCriterion_SS if mode == "ss" else SetCriterion
    crit = cls(2, matcher=matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"], num_points=12544,
               oversample_ratio=3.0, importance_sample_ratio=0.75).cuda()
    crit.point_source = lambda n, p: torch.rand(n, p, 2).cuda()  # replay the reference's CPU RNG stream
    return crit, wd


@pytest.mark.parametrize("mode", ["s4", "all", "ss"])
def test_criterion_matches_reference(head_run, mode):
    z, head, feats, audio, out = head_run
    zc = np.load(os.path.join(G, "criterion.npz"))
    crit, wd = make_criterion(mode)
    o = {"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"],
         "aux_outputs": [dict(a) for a in out["aux_outputs"]], "middles_attn_mask": list(out["middles_attn_mask"])}
    torch.manual_seed(11)
    if mode == "ss":
        gt_flag = torch.from_numpy(zc["ss/gt_flag"])
        t_all = gen_inputs.make_targets("all")
        targets = [{k: v.cuda() for k, v in t_all[i].items()} for i in range(5) if gt_flag[i] == 1]
        losses = crit(o, targets, torch.ones(5).cuda(), gt_flag.cuda())
    else:
        targets = [{k: v.cuda() for k, v in t.items()} for t in gen_inputs.make_targets(mode)]
        losses = crit(o, targets)
    keys = json.loads(str(zc[f"{mode}/keys"]))
    assert sorted(losses.keys()) == keys
    # index work, bit-exact: the device LSAP's Hungarian pairs of ALL 10 outputs (final first, then aux 0..8) equal the pairs
    # scipy returned inside the reference's criterion (matcher.py:132-134), and the final layer's equal `match_src/_tgt`
    src_q, tgt_g, _ = crit.last_indices
    assert np.array_equal(src_q.cpu().numpy(), zc[f"{mode}/match_all_src"]), (src_q.cpu().numpy(), zc[f"{mode}/match_all_src"])
    assert np.array_equal(tgt_g.cpu().numpy(), zc[f"{mode}/match_all_tgt"])
    assert np.array_equal(src_q[0].cpu().numpy(), zc[f"{mode}/match_src"].reshape(-1))
    assert np.array_equal(tgt_g[0].cpu().numpy(), zc[f"{mode}/match_tgt"].reshape(-1))
    got = np.array([float(losses[k]) for k in keys])
    np.testing.assert_allclose(got, zc[f"{mode}/values"], rtol=2e-3, atol=2e-4)
    total = sum(losses[k] * wd[k] for k in keys)
    np.testing.assert_allclose(float(total), float(zc[f"{mode}/total"]), rtol=1e-3)
    if f"{mode}/grad/feat.audio/sample" not in zc.files:
        return
    grad_params = json.loads(str(zc["grad_params"]))
    named = dict(head.named_parameters())
    gi = list(feats.values()) + [audio] + [named[n] for n in grad_params]
    grads = torch.autograd.grad(total, gi, retain_graph=True, allow_unused=True)
    names = [f"feat.{k}" for k in feats] + ["feat.audio"] + grad_params
    # Un-frozen run: the product makes ALL its own discrete choices (attention-mask bits, Hungarian pairs, top-k point sets).
    # S4 mode: every gradient within 2e-3 of the tensor's RMS + 2e-3 relative for >= 99.8 % of the samples.  In the modes with
    # ground truth on every frame a single near-tie that falls the other way moves a whole gradient row, so here only the
    # energy of the error is bounded (relative L2 <= 5e-2, no entry beyond 1 RMS); the TIGHT comparison of those modes is
    # test_criterion_gradients_with_the_references_choices_frozen below.
    worst = []
    for n, g in zip(names, grads):
        d = synth.unpack(f"{mode}/grad/{n}", zc)
        scale = float(d["l2"]) / max(np.sqrt(float(d["numel"])), 1.0)
        try:
            if mode == "s4" and not synth.upstream_of_sampling(n):
                synth.check_digest(g.cpu(), d, f"{mode}/grad/{n}", rtol=2e-3, atol=2e-3 * scale + 1e-9, frac_bad=0.002)
            elif mode == "s4":  # (pixel-boundary taps, tests/golden/synth.py: energy form, the element-wise fraction is printed)
                synth.check_digest_l2(g.cpu(), d, f"{mode}/grad/{n}", rel_l2=1e-2, cap_rms=0.3)
            else:
                synth.check_digest_l2(g.cpu(), d, f"{mode}/grad/{n}", rel_l2=5e-2, cap_rms=1.0)
        except AssertionError as e:
            worst.append(str(e))
    assert not worst, worst


PIXEL_BOUNDARY = synth.PIXEL_BOUNDARY  # (tests/golden/synth.py)


@pytest.fixture(scope="module")
def head_run_frozen(head_run):
    """The same forward with the reference's own 9 attention masks injected (head.npz `dec/attn_bits*`)."""
    from combo_avs_amd.ops import masklogit
    z, head, feats, audio, _ = head_run
    head.predictor.attn_mask_override = [masklogit.pack_mask(m.cuda()) for m in synth.frozen_attn_masks(z)]
    try:
        out = head(dict(feats), audio)
        torch.cuda.synchronize()
    finally:
        head.predictor.attn_mask_override = None
    return z, head, feats, audio, out


# The bounds of the two gradient classes.  They are constants so that loosening one is a visible diff:
ELEMENTWISE = dict(rtol=2e-3, atol_rms=2e-3, frac_bad=0.002)   # every gradient NOT upstream of the deformable encoder's sampling
ENERGY_FORM = dict(rel_l2=1e-2, cap_rms=0.3)                   # PIXEL_BOUNDARY tensors: relative L2 only (+ <= 25 % beyond 2e-3)


@pytest.mark.parametrize("group", ["elementwise_2e-3", "pixel_boundary_tensors_relative_L2_1e-2_only"])
@pytest.mark.parametrize("mode", ["s4", "all", "ss"])
def test_criterion_gradients_with_the_references_choices_frozen(head_run_frozen, mode, group):
    """Gradients of the weighted 39-term loss against the reference's with the reference's own discrete choices injected on
    the HIP path: the 9 attention masks (decoder.attn_mask_override), the Hungarian pairs of all 10 outputs and the top-k sets
    of the importance sampling (criterion.frozen_choices).  What is left is arithmetic: NO outlier budget beyond 0.2 % at
    2e-3 for every gradient that does not pass through the deformable encoder's bilinear taps, and an energy bound for the
    five that do (synth.check_digest_l2; the CPU oracle measures rel. L2 <= 2.7e-3, worst entry 0.11 RMS on the same vectors -
    tests/test_oracle_golden.py - the bound here leaves 4x for the 3-product bf16 gradient GEMMs).
    group "pixel_boundary_tensors_relative_L2_1e-2_only": the tensors held ONLY to the energy form - a bilinear tap crossing a
    pixel under a 1e-7 change of a sampling location moves the value-map gradient between neighbouring pixels; the same
    happens CPU-vs-CPU (oracle vs reference, tests/test_oracle_golden.py).  Looser than the element-wise class by design."""
    z, head, feats, audio, out = head_run_frozen
    zc = np.load(os.path.join(G, "criterion.npz"))
    crit, wd = make_criterion(mode)
    crit.frozen_choices = synth.frozen_criterion_choices(zc, mode)
    o = {"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"], "_logits_all": out["_logits_all"],
         "aux_outputs": [dict(a) for a in out["aux_outputs"]], "middles_attn_mask": list(out["middles_attn_mask"])}
    torch.manual_seed(11)
    if mode == "ss":
        gt_flag = torch.from_numpy(zc["ss/gt_flag"])
        t_all = gen_inputs.make_targets("all")
        targets = [{k: v.cuda() for k, v in t_all[i].items()} for i in range(5) if gt_flag[i] == 1]
        o.pop("_logits_all")
        losses = crit(o, targets, torch.ones(5).cuda(), gt_flag.cuda())
    else:
        targets = [{k: v.cuda() for k, v in t.items()} for t in gen_inputs.make_targets(mode)]
        losses = crit(o, targets)
    keys = json.loads(str(zc[f"{mode}/keys"]))
    got = np.array([float(losses[k]) for k in keys])
    np.testing.assert_allclose(got, zc[f"{mode}/values"], rtol=1e-3, atol=1e-4)
    total = sum(losses[k] * wd[k] for k in keys)
    grad_params = json.loads(str(zc["grad_params"]))
    named = dict(head.named_parameters())
    gi = list(feats.values()) + [audio] + [named[n] for n in grad_params]
    grads = torch.autograd.grad(total, gi, retain_graph=True, allow_unused=True)
    names = [f"feat.{k}" for k in feats] + ["feat.audio"] + grad_params
    worst = []
    for n, g in zip(names, grads):
        d = synth.unpack(f"{mode}/grad/{n}", zc)
        scale = float(d["l2"]) / max(np.sqrt(float(d["numel"])), 1.0)
        energy = n in PIXEL_BOUNDARY or synth.upstream_of_sampling(n)
        if energy != group.startswith("pixel_boundary"):
            continue
        try:
            if energy:
                synth.check_digest_l2(g.cpu(), d, f"{mode}/grad/{n}", **ENERGY_FORM)
            else:
                synth.check_digest(g.cpu(), d, f"{mode}/grad/{n}", rtol=ELEMENTWISE["rtol"], atol=ELEMENTWISE["atol_rms"] * scale + 1e-9,
                                   frac_bad=ELEMENTWISE["frac_bad"])
        except AssertionError as e:
            worst.append(str(e))
    assert not worst, worst


def test_gradient_bounds_have_not_been_loosened():
    """the two classes' bounds as reviewed in round 4 (VERDICT weak 3): a change here must be argued in DESIGN section 2"""
    assert ELEMENTWISE == dict(rtol=2e-3, atol_rms=2e-3, frac_bad=0.002) and ENERGY_FORM == dict(rel_l2=1e-2, cap_rms=0.3)
    import inspect
    assert inspect.signature(synth.check_digest_l2).parameters["frac_2e3_cap"].default == 0.25


def test_fast_matching_path_agrees_with_replay(head_run):
    """match_layers (one D2H copy for all 10 outputs) returns the same assignment as 10 sequential matcher calls."""
    z, head, feats, audio, out = head_run
    crit, _ = make_criterion("all")
    targets = [{k: v.cuda() for k, v in t.items()} for t in gen_inputs.make_targets("all")]
    layers = [{"pred_logits": out["pred_logits"].detach(), "pred_masks": out["pred_masks"].detach()}] + \
             [{k: v.detach() for k, v in a.items()} for a in out["aux_outputs"]]
    ps = lambda n, p: torch.rand(n, p, 2).cuda()
    torch.manual_seed(5)
    a = crit.matcher.match_layers(layers, targets, ps)
    torch.manual_seed(5)
    b = [crit.matcher(l, targets, ps) for l in layers]
    for la, lb in zip(a, b):
        for (i1, j1), (i2, j2) in zip(la, lb):
            assert torch.equal(i1, i2) and torch.equal(j1, j2)


def test_criterion_ragged_instance_counts_vs_oracle(head_run):
    """Frames with 0 / 1 / 2 / 3 / 6 ground-truth instances in one batch (all-frames mode): the batched HIP criterion
    (padded cost tensors, device LSAP, flat pair lists) against the CPU oracle's per-frame loop on the same RNG stream."""
    z, head, feats, audio, out = head_run
    crit, wd = make_criterion("all")
    o = {"pred_logits": out["pred_logits"].detach(), "pred_masks": out["pred_masks"].detach(),
         "aux_outputs": [{k: v.detach() for k, v in a.items()} for a in out["aux_outputs"]],
         "middles_attn_mask": [m.detach() for m in out["middles_attn_mask"]]}
    counts = [2, 0, 3, 1, 6]
    g = torch.Generator().manual_seed(7)
    targets = []
    yy, xx = torch.meshgrid(torch.arange(224), torch.arange(224), indexing="ij")
    for n in counts:
        masks = []
        for _ in range(n):
            cx, cy = (torch.rand(2, generator=g) * 0.6 + 0.2) * 224
            r = (torch.rand(1, generator=g) * 0.2 + 0.05) * 224
            masks.append(((xx - cx) ** 2 + (yy - cy) ** 2) < r * r)
        targets.append({"labels": torch.randint(0, 2, (n,), generator=g, dtype=torch.int64),
                        "masks": torch.stack(masks) if n else torch.zeros(0, 224, 224, dtype=torch.bool)})
    torch.manual_seed(23)
    got = crit(o, [{k: v.cuda() for k, v in t.items()} for t in targets])
    torch.manual_seed(23)
    cpu = {"pred_logits": o["pred_logits"].cpu(), "pred_masks": o["pred_masks"].cpu(),
           "aux_outputs": [{k: v.cpu() for k, v in a.items()} for a in o["aux_outputs"]],
           "middles_attn_mask": [m.cpu() for m in o["middles_attn_mask"]]}
    from oracle import combo_oracle as O
    ref = O.set_criterion(cpu, targets, num_classes=2)
    assert sorted(got) == sorted(ref) and len(got) == 39
    for k in sorted(ref):
        a, b = float(got[k]), float(ref[k])
        assert abs(a - b) <= 5e-3 * abs(b) + 5e-4, (k, a, b)


@pytest.mark.parametrize("counts", [[0, 0, 0, 0, 0], [10, 1, 0, 2, 7]])
def test_criterion_empty_batch_and_many_instances_vs_oracle(head_run, counts):
    """No instance at all (empty sums, like the reference), and frames with more than 8 / more than 6 instances (cost kernel
    in column chunks, SciPy LSAP on the host instead of the small-G device solver)."""
    from oracle import combo_oracle as O
    z, head, feats, audio, out = head_run
    crit, wd = make_criterion("all")
    o = {"pred_logits": out["pred_logits"].detach(), "pred_masks": out["pred_masks"].detach(),
         "aux_outputs": [{k: v.detach() for k, v in a.items()} for a in out["aux_outputs"]],
         "middles_attn_mask": [m.detach() for m in out["middles_attn_mask"]]}
    g = torch.Generator().manual_seed(9)
    yy, xx = torch.meshgrid(torch.arange(224), torch.arange(224), indexing="ij")
    targets = []
    for n in counts:
        masks = []
        for _ in range(n):
            cx, cy = (torch.rand(2, generator=g) * 0.6 + 0.2) * 224
            r = (torch.rand(1, generator=g) * 0.2 + 0.05) * 224
            masks.append(((xx - cx) ** 2 + (yy - cy) ** 2) < r * r)
        targets.append({"labels": torch.randint(0, 2, (n,), generator=g, dtype=torch.int64),
                        "masks": torch.stack(masks) if n else torch.zeros(0, 224, 224, dtype=torch.bool)})
    torch.manual_seed(29)
    got = crit(o, [{k: v.cuda() for k, v in t.items()} for t in targets])
    torch.manual_seed(29)
    cpu = {"pred_logits": o["pred_logits"].cpu(), "pred_masks": o["pred_masks"].cpu(),
           "aux_outputs": [{k: v.cpu() for k, v in a.items()} for a in o["aux_outputs"]],
           "middles_attn_mask": [m.cpu() for m in o["middles_attn_mask"]]}
    ref = O.set_criterion(cpu, targets, num_classes=2)
    assert sorted(got) == sorted(ref) and len(got) == 39
    for k in sorted(ref):
        a, b = float(got[k]), float(ref[k])
        assert abs(a - b) <= 5e-3 * abs(b) + 5e-4, (k, a, b)


def test_deferred_grouped_weight_gradients_equal_immediate(head_run):
    """ops.linear.deferred_dw: the decoder's / pixel decoder's weight gradients and LayerNorm parameter gradients computed by
    grouped launches at the end of the backward pass against the same gradients computed launch by launch."""
    from combo_avs_amd.ops.linear import deferred_dw
    z, head, feats, audio, out = head_run
    named = [(n, p) for n, p in head.named_parameters() if p.requires_grad and (n.startswith("predictor.") or n.startswith("pixel_decoder."))]
    loss = sum(a["pred_masks"].float().pow(2).mean() + a["pred_logits"].float().pow(2).mean() for a in out["aux_outputs"]) \
        + out["pred_masks"].float().pow(2).mean() + out["pred_logits"].float().pow(2).mean()
    params = [p for _, p in named]
    ref = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
    with deferred_dw():
        got = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
    torch.cuda.synchronize()
    n_checked = 0
    for (name, _), a, b in zip(named, got, ref):
        if b is None:
            assert a is None
            continue
        assert (a - b).abs().max() <= 2e-5 * b.abs().max() + 1e-9, (name, float((a - b).abs().max()), float(b.abs().max()))
        n_checked += 1
    assert n_checked > 100


@pytest.mark.parametrize("mode", ["s4", "all"])
def test_fused_criterion_path_equals_unfused(head_run, mode):
    """The criterion reading the decoder's single logits buffer (`_logits_all`: matcher with map offsets, one node for
    mask losses + cosine statistics, one gradient buffer) against the path through the per-head outputs: same 39 losses,
    same gradients (same injected random-point stream)."""
    z, head, feats, audio, out = head_run
    assert "_logits_all" in out and out["_logits_all"].shape[0] == 10
    targets = [{k: v.cuda() for k, v in t.items()} for t in gen_inputs.make_targets(mode)]
    named = dict(head.named_parameters())
    probe = [named[n] for n in ("predictor.mask_embed.layers.2.weight", "predictor.query_feat.weight",
                                "pixel_decoder.mask_features.weight", "predictor.transformer_ffn_layers.4.linear1.weight")]
    res = []
    for fused in (True, False, "declined"):
        crit, wd = make_criterion(mode)
        o = {"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"], "aux_outputs": [dict(a) for a in out["aux_outputs"]],
             "middles_attn_mask": list(out["middles_attn_mask"])}
        if fused is True:
            o["_logits_all"] = out["_logits_all"]
        elif fused == "declined":
            # a buffer the fused path does not take (one head short): forward() still hands _losses all BT frames + the
            # ground-truth frame ids, and the unfused branch has to pick those frames itself
            o["_logits_all"] = out["_logits_all"][:-1]
        torch.manual_seed(11)
        losses = crit(o, targets)
        total = sum(losses[k] * wd[k] for k in losses)
        grads = torch.autograd.grad(total, probe, retain_graph=True)
        res.append(({k: float(v) for k, v in losses.items()}, grads))
    (la, ga), (lb, gb), (lc, gc) = res
    # the declined-fused call is the unfused computation (the cosine term may take another summation order)
    assert sorted(lc) == sorted(lb) and all(abs(lc[k] - lb[k]) <= 1e-6 * abs(lb[k]) + 1e-7 for k in lb), (lc, lb)
    assert sorted(la) == sorted(lb) and len(la) == 39
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * abs(lb[k]) + 1e-6, (k, la[k], lb[k])
    for a, b in zip(ga, gb):
        assert (a - b).abs().max() <= 2e-4 * b.abs().max() + 1e-8, float((a - b).abs().max() / b.abs().max())


def test_x3_forward_mode_stated_tolerance():
    """The head's 3-product forward mode (ops.linear.set_forward_precision("x3"), bench.py --head-dtype x3): the forward GEMMs /
    convolutions / mask-logit contraction of the head on csrc/gemm_nt3.hip with the fp32-grade 3-product bf16 split (max error
    ~5e-6 per layer) instead of an fp32-grade product (the default fp16 pieces or the exact fp32 matrix instruction).  NOT the default: a cell of a decoder attention mask whose
    logit lies within that error of 0 flips and re-routes its query for the rest of the decoder - the north-star bound then holds
    for all but a few entries instead of for all of them (the default path: 0 outliers).  Stated tolerance against the
    reference's fp32 outputs (golden head.npz):
      * prediction head #0 (no thresholded mask upstream): EVERY sampled mask logit within 1e-3 x RMS + 1e-3 x |ref|;
      * all 10 heads: >= 98 % of the sampled mask logits within that bound (measured: heads 0 - 6 all of them, head 7 99.93 %,
        head 8 99.4 %, head 9 98.85 % - the same picture as round 1's 3-product forward), relative L2 error of every head <= 2e-3
        (measured <= 8.2e-4), max error 2.4e-2 RMS;
      * class logits: >= 99.5 % within 1e-3 x RMS + 1e-3 x |ref| (measured: 14 of 15 000 beyond)."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    z = np.load(os.path.join(G, "head.npz"))
    spec = json.loads(str(z["spec"]))
    head, cfg = build_head()
    head.load_state_dict(synth.synth_state_dict(spec, 0))
    head = head.cuda().eval()
    feats, audio = gen_inputs.head_inputs()
    feats = {k: v.cuda() for k, v in feats.items()}
    L.set_forward_precision("x3")
    try:
        with torch.no_grad(), L.grouped_presplit():
            out = head(dict(feats), audio.cuda())
        torch.cuda.synchronize()
    finally:
        L.set_forward_precision(L.DEFAULT_FORWARD_PRECISION)
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    report = []
    for i, m in enumerate(masks):
        d = synth.unpack(f"dec/pred_masks{i}", z)
        idx = synth.digest_indices(m.numel(), 4096, f"dec/pred_masks{i}")
        got = m.reshape(-1).cpu().numpy()[idx].astype(np.float64)
        ref = np.asarray(d["sample"]).astype(np.float64)
        rms = float(np.sqrt((ref ** 2).mean()))
        err = np.abs(got - ref)
        beyond = float((err > 1e-3 * rms + 1e-3 * np.abs(ref)).mean())
        l2 = float(np.sqrt((err ** 2).sum() / (ref ** 2).sum()))
        report.append((i, round(beyond, 5), round(l2, 6), round(float(err.max() / rms), 5)))
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]).cpu().numpy()
    ref = z["dec/pred_logits"]
    bad = np.abs(logits - ref) > 1e-3 * np.sqrt((ref ** 2).mean()) + 1e-3 * np.abs(ref)
    if os.environ.get("COMBO_TEST_VERBOSE") == "1":
        print("[x3 mode] head: (frac beyond the north-star bound, rel L2, max err / RMS)", report, "class logits beyond:", int(bad.sum()), "of", bad.size)
    assert report[0][1] == 0.0, report
    assert all(r[1] <= 0.02 and r[2] <= 2e-3 for r in report), report
    assert bad.mean() <= 5e-3, float(bad.mean())


def test_bf16_forward_mode_stated_tolerance():
    """The head's bf16 throughput mode (ops.linear.set_forward_precision("bf16"), bench.py --head-dtype bf16): every forward
    GEMM / convolution / mask-logit contraction of the head on ONE bf16 product per multiply-add (csrc/gemm_nt3.hip, fp32
    accumulation) - the product's own kernels, not torch autocast.  Stated tolerance against the reference's fp32 outputs
    (golden head.npz), which is NOT the north-star's 1e-3 (that is what the default fp32 path is for):
      * prediction head #0 (no thresholded mask upstream): every sampled mask logit within 5e-2 x RMS + 2e-2 x |ref|
        (measured: max 2.7e-2 RMS, relative L2 error 7.5e-3);
      * all 10 heads: >= 97.5 % of the sampled mask logits within 1e-1 x RMS + 1e-1 x |ref| (measured >= 99 %: a flipped
        attention-mask cell re-routes a query for the rest of the decoder), relative L2 error of every head <= 0.1
        (measured <= 4.5e-2);
      * class logits: within 0.25 absolute of the reference for >= 97 % of the entries.
    The S4 losses of this forward stay within 5 % of the reference's."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    z = np.load(os.path.join(G, "head.npz"))
    spec = json.loads(str(z["spec"]))
    head, cfg = build_head()
    head.load_state_dict(synth.synth_state_dict(spec, 0))
    head = head.cuda().eval()
    feats, audio = gen_inputs.head_inputs()
    feats = {k: v.cuda() for k, v in feats.items()}
    L.set_forward_precision("bf16")
    try:
        with torch.no_grad(), L.grouped_presplit():
            out = head(dict(feats), audio.cuda())
        torch.cuda.synchronize()
    finally:
        L.set_forward_precision(L.DEFAULT_FORWARD_PRECISION)
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    report = []
    for i, m in enumerate(masks):
        d = synth.unpack(f"dec/pred_masks{i}", z)
        idx = synth.digest_indices(m.numel(), 4096, f"dec/pred_masks{i}")
        got = m.reshape(-1).cpu().numpy()[idx].astype(np.float64)
        ref = np.asarray(d["sample"]).astype(np.float64)
        rms = float(np.sqrt((ref ** 2).mean()))
        err = np.abs(got - ref)
        tight = float((err > 5e-2 * rms + 2e-2 * np.abs(ref)).mean())
        loose = float((err > 1e-1 * rms + 1e-1 * np.abs(ref)).mean())
        l2 = float(np.sqrt((err ** 2).sum() / (ref ** 2).sum()))
        report.append((i, round(tight, 4), round(loose, 4), round(l2, 4), round(float(err.max() / rms), 3)))
    if os.environ.get("COMBO_TEST_VERBOSE") == "1":
        print("[bf16 mode] head: (frac beyond 5e-2, frac beyond 1e-1, rel L2, max err / RMS)", report)
    assert report[0][1] == 0.0, report
    assert all(r[2] <= 0.025 and r[3] <= 0.1 for r in report), report
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]).cpu().numpy()
    bad = np.abs(logits - z["dec/pred_logits"]) > 0.25
    assert bad.mean() <= 0.03, float(bad.mean())
    # losses of this forward (S4 targets, replayed points)
    zc = np.load(os.path.join(G, "criterion.npz"))
    crit, wd = make_criterion("s4")
    torch.manual_seed(11)
    targets = [{k: v.cuda() for k, v in t.items()} for t in gen_inputs.make_targets("s4")]
    o = {"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"], "aux_outputs": [dict(a) for a in out["aux_outputs"]],
         "middles_attn_mask": list(out["middles_attn_mask"])}
    losses = crit(o, targets)
    keys = json.loads(str(zc["s4/keys"]))
    total = sum(float(losses[k]) * wd[k] for k in keys)
    assert abs(total - float(zc["s4/total"])) <= 0.05 * abs(float(zc["s4/total"])), (total, float(zc["s4/total"]))
