"""GPU parity of the product head + criterion at the geometry of BASELINE configs[3] / configs[4] against golden vectors that the
REFERENCE's own head + criterion produced at that geometry (tests/golden/gen_golden_cfg34.py -> head_{case}.npz /
criterion_{case}.npz; round 6 - until round 5 these configs ran at full size but were judged by finite-loss / parameter-motion /
adjointness properties only):

  ms3_t10   PVTv2-B5 widths (64 / 128 / 320 / 512), one clip of 10 frames at 224 x 224, NUM_FRAMES = 10, K = 2, SetCriterion with
            ground truth on every frame (configs/avs_ms3/COMBO_PVTV2B5_bs8_20k.yaml + MODEL.FUSE_CONFIG.NUM_FRAMES 10)
  avss_512  the same widths at 512 x 512: 128 x 128 mask features, S = 5376 encoder tokens (MSDeformAttn's generic forward +
            windowed backward), K = 71, SetCriterion_SS with the AVSS flag tensors: a v1s clip (ground truth on its first frame)
            and a v1m clip (on all five) - configs/avs_ss/COMBO_PVTV2B5_bs8_90k.yaml

Bounds: the north-star's 1e-3 * RMS(head) + 1e-3 * |ref| on class and mask logits of ALL 10 prediction heads with no outlier
budget (the reference's attention-mask bits injected, as in tests/test_model_gpu.py: a logit within round-off of 0 must not
re-route a query); the un-injected run is held to the same bound on head 0 and its flipped cells are counted; Hungarian pairs
bit-for-bit; the 39 losses; gradient digests in the two classes of tests/test_head_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

import gen_inputs
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CASES = {
    "ms3_t10": dict(yaml="configs/avs_ms3/COMBO_PVTV2B5_bs8_20k.yaml", opts=("MODEL.FUSE_CONFIG.NUM_FRAMES", 10),
                    levels=((7, 7), (14, 14), (28, 28))),
    "avss_512": dict(yaml="configs/avs_ss/COMBO_PVTV2B5_bs8_90k.yaml", opts=(), levels=((16, 16), (32, 32), (64, 64))),
}
ELEMENTWISE = dict(rtol=2e-3, atol_rms=2e-3, frac_bad=0.005)  # (0.5 %, not test_head_gpu.py's 0.2 %: see test_oracle_golden_cfg34.py)
ENERGY_FORM = dict(rel_l2=1e-2, cap_rms=0.3)


def load_case(case):
    z = np.load(os.path.join(G, f"head_{case}.npz"), allow_pickle=False)
    zc = np.load(os.path.join(G, f"criterion_{case}.npz"), allow_pickle=False)
    return z, zc, json.loads(str(z["case"]))


def build_head(case):
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import combo_cfg
    from combo_avs_amd.meta_arch import build_sem_seg_head
    from combo_avs_amd.registry import ShapeSpec
    cfg = combo_cfg(os.path.join(ROOT, CASES[case]["yaml"]), opts=CASES[case]["opts"])
    shapes = {f"res{i + 2}": ShapeSpec(channels=c, stride=4 * 2 ** i) for i, c in enumerate((64, 128, 320, 512))}
    return build_sem_seg_head(cfg, shapes), cfg


def targets_of(case, c, device="cuda"):
    t_all = (gen_inputs.make_targets("all", bt=c["bt"], size=c["size"]) if c["K"] == 2
             else gen_inputs.make_targets_k(c["bt"], c["size"], c["K"], case))
    if c["crit"] == "ss":
        t_all = [t_all[i] for i in range(c["bt"]) if c["gt_flag"][i] == 1]
    return [{k: v.to(device) for k, v in t.items()} for t in t_all]


@pytest.fixture(scope="module", params=list(CASES))
def run(request):
    from combo_avs_amd.ops import masklogit
    case = request.param
    z, zc, c = load_case(case)
    spec = json.loads(str(z["spec"]))
    head, cfg = build_head(case)
    ours = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    assert ours == {k: tuple(s) for k, s in spec}, "state-dict surface differs from the reference's"
    head.load_state_dict(synth.synth_state_dict(spec, 0))
    head = head.cuda().eval()
    feats, audio = gen_inputs.head_inputs(bt=c["bt"], hw=c["hw"], channels=tuple(c["channels"]), tag=f"feat.{case}")
    feats = {k: v.cuda().requires_grad_(True) for k, v in feats.items()}
    audio = audio.cuda().requires_grad_(True)
    with torch.no_grad():
        free = head({k: v.detach() for k, v in feats.items()}, audio.detach())  # the product's own attention masks
        free = {"masks": [a["pred_masks"] for a in free["aux_outputs"]] + [free["pred_masks"]],
                "logits": [a["pred_logits"] for a in free["aux_outputs"]] + [free["pred_logits"]]}
    ref_masks = synth.frozen_attn_masks(z, bt=c["bt"], sizes=CASES[case]["levels"])
    head.predictor.attn_mask_override = [masklogit.pack_mask(m.cuda()) for m in ref_masks]
    try:
        out = head(dict(feats), audio)
        torch.cuda.synchronize()
    finally:
        head.predictor.attn_mask_override = None
    return case, z, zc, c, head, feats, audio, out, free, ref_masks


def _heads(out):
    return ([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]],
            [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]])


def test_all_ten_heads_match_the_reference(run):
    case, z, zc, c, head, feats, audio, out, free, ref_masks = run
    logits, masks = _heads(out)
    assert tuple(torch.stack(logits).shape) == tuple(z["dec/logits_shape"])
    assert tuple(out["middles_attn_mask"][0].shape) == tuple(z["dec/middle_shape"]) and len(out["middles_attn_mask"]) == 9
    worst = 0.0
    for i, (lg, m) in enumerate(zip(logits, masks)):
        for nm, t in ((f"dec/pred_logits{i}", lg), (f"dec/pred_masks{i}", m)):
            d = synth.unpack(nm, z)
            rms = float(d["l2"]) / np.sqrt(float(d["numel"]))
            worst = max(worst, synth.check_digest(t.detach().cpu(), d, nm, rtol=1e-3, atol=1e-3 * rms, k=8192, frac_bad=0.0) / rms)
    print(f"[{case}] class + mask logits of all 10 heads within 1e-3 RMS + 1e-3 |ref| of the reference's (8 192 samples per tensor, "
          f"0 outliers); worst sampled error {worst:.2e} RMS")


def test_own_attention_masks_and_first_head(run):
    """the un-injected forward: head 0 (no earlier mask involved) at the same bound; the product's thresholded masks against the
    reference's bit rows - cells may differ only where the down-sampled logit is within round-off of the threshold"""
    case, z, zc, c, head, feats, audio, out, free, ref_masks = run
    for nm, t in (("dec/pred_logits0", free["logits"][0]), ("dec/pred_masks0", free["masks"][0])):
        d = synth.unpack(nm, z)
        rms = float(d["l2"]) / np.sqrt(float(d["numel"]))
        synth.check_digest(t.cpu(), d, nm, rtol=1e-3, atol=1e-3 * rms, k=8192, frac_bad=0.0)
    flips, far = [], 0
    for i in range(9):
        m = free["masks"][i]
        d = synth.unpack(f"dec/pred_masks{i}", z)
        rms = float(d["l2"]) / np.sqrt(float(d["numel"]))
        tgt = CASES[case]["levels"][i % 3]
        down = torch.nn.functional.interpolate(m, size=tgt, mode="bilinear", align_corners=False).flatten(2)
        blocked = (down.sigmoid() < 0.5).cpu()
        near = (down.abs() < 1e-5 * rms).cpu()
        flips.append(int((blocked != ref_masks[i]).sum()))
        if i == 0:
            far = int(((blocked != ref_masks[0]) & ~near).sum())
    print(f"[{case}] un-injected run: attention-mask cells that differ from the reference's, heads 0..8: {flips} of "
          f"{[int(m.numel()) for m in ref_masks]}")
    assert far == 0 and flips[0] <= 4, (far, flips)
    # later heads inherit earlier flips (a re-routed query): bounded as a share of the cells, reported above
    assert max(f / m.numel() for f, m in zip(flips, ref_masks)) < 2e-3, flips


def test_intermediates_match_the_reference(run):
    case, z, zc, c, head, feats, audio, out, free, ref_masks = run
    with torch.no_grad():
        mf, _, ms = head.pixel_decoder.forward_features({k: v.detach() for k, v in feats.items()})
        synth.check_digest(mf.cpu(), synth.unpack("pd/mask_features", z), "pd/mask_features", 1e-3, 2e-4)
        for i, m in enumerate(ms):
            synth.check_digest(m.cpu(), synth.unpack(f"pd/ms{i}", z), f"pd/ms{i}", 1e-3, 2e-4)
        fused = head.fusion_module({"res2": mf}, audio.detach())
        synth.check_digest(fused["visual"]["res2"].contiguous().cpu(), synth.unpack("fuse/visual", z), "fuse/visual", 1e-3, 2e-4)
        np.testing.assert_allclose(fused["audio"].cpu().numpy(), z["fuse/audio"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(head.audio_transformation(fused["audio"]).cpu().numpy(), z["fuse/audio256"], rtol=1e-3, atol=2e-4)


def make_criterion(c):
    from combo_avs_amd.modeling.criterion import SetCriterion, SetCriterion_SS
    from combo_avs_amd.modeling.matcher import HungarianMatcher
    w = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 5.0, "loss_cosine": 10.0}
    wd = dict(w)
    for i in range(9):
        wd.update({f"{k}_{i}": v for k, v in w.items()})
    matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=12544)
    cls = SetCriterion_SS if c["crit"] == "ss" else SetCriterion
    crit = cls(c["K"], matcher=matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"], num_points=12544,
               oversample_ratio=3.0, importance_sample_ratio=0.75).cuda()
    crit.point_source = lambda n, p: torch.rand(n, p, 2).cuda()  # replay the reference's CPU RNG stream
    return crit, wd


def _losses(c, crit, out, targets):
    o = {"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"],
         "aux_outputs": [dict(a) for a in out["aux_outputs"]], "middles_attn_mask": list(out["middles_attn_mask"])}
    torch.manual_seed(11)
    if c["crit"] == "ss":
        return crit(o, targets, torch.ones(c["bt"]).cuda(), torch.tensor(c["gt_flag"], dtype=torch.float32).cuda())
    return crit(o, targets)


def test_criterion_matches_the_reference(run):
    """the product's own discrete choices on the reference's RNG stream: Hungarian pairs of all 10 outputs bit-for-bit (device
    LSAP, K = 71 classes / 1 - 4 instances per frame in the AVSS case), the 39 losses (criterion.py:233-287,
    criterion_ss.py:238-289; cosine grouping in fives: criterion.py:284)"""
    case, z, zc, c, head, feats, audio, out, free, ref_masks = run
    crit, wd = make_criterion(c)
    targets = targets_of(case, c)
    assert [t["labels"].tolist() for t in targets] == json.loads(str(zc["labels"]))
    losses = _losses(c, crit, out, targets)
    keys = json.loads(str(zc["keys"]))
    assert sorted(losses.keys()) == keys and len(keys) == 39
    src_q, tgt_g, _ = crit.last_indices
    assert np.array_equal(src_q.cpu().numpy(), zc["match_all_src"]), (src_q.cpu().numpy(), zc["match_all_src"])
    assert np.array_equal(tgt_g.cpu().numpy(), zc["match_all_tgt"])
    got = np.array([float(losses[k]) for k in keys])
    np.testing.assert_allclose(got, zc["values"], rtol=2e-3, atol=2e-4)
    total = sum(losses[k] * wd[k] for k in keys)
    np.testing.assert_allclose(float(total), float(zc["total"]), rtol=1e-3)
    print(f"[{case}] 39 losses within 2e-3 of the reference's, total {float(total):.4f} vs {float(zc['total']):.4f}")


@pytest.mark.parametrize("group", ["elementwise_2e-3", "upstream_of_sampling_relative_L2_1e-2_only"])
def test_gradients_with_the_references_choices_frozen(run, group):
    case, z, zc, c, head, feats, audio, out, free, ref_masks = run
    crit, wd = make_criterion(c)
    n_over = 37632
    topk = torch.from_numpy(np.unpackbits(zc["topk_bits"], axis=2)[:, :, :n_over].astype(bool))
    crit.frozen_choices = {"match_src": torch.from_numpy(zc["match_all_src"]), "match_tgt": torch.from_numpy(zc["match_all_tgt"]), "topk": topk}
    losses = _losses(c, crit, out, targets_of(case, c))
    keys = json.loads(str(zc["keys"]))
    got = np.array([float(losses[k]) for k in keys])
    np.testing.assert_allclose(got, zc["values"], rtol=1e-3, atol=1e-4)
    total = sum(losses[k] * wd[k] for k in keys)
    grad_params = json.loads(str(zc["grad_params"]))
    named = dict(head.named_parameters())
    gi = list(feats.values()) + [audio] + [named[n] for n in grad_params]
    grads = torch.autograd.grad(total, gi, retain_graph=True, allow_unused=True)
    names = [f"feat.{k}" for k in feats] + ["feat.audio"] + grad_params
    worst = []
    for n, g in zip(names, grads):
        d = synth.unpack(f"grad/{n}", zc)
        scale = float(d["l2"]) / max(np.sqrt(float(d["numel"])), 1.0)
        energy = synth.upstream_of_sampling(n)
        if energy != group.startswith("upstream"):
            continue
        try:
            if energy:
                synth.check_digest_l2(g.cpu(), d, f"{case}/grad/{n}", k=2048, **ENERGY_FORM)
            else:
                synth.check_digest(g.cpu(), d, f"{case}/grad/{n}", rtol=ELEMENTWISE["rtol"], atol=ELEMENTWISE["atol_rms"] * scale + 1e-9,
                                   k=2048, frac_bad=ELEMENTWISE["frac_bad"])
        except AssertionError as e:
            worst.append(str(e))
    assert not worst, worst
