"""CPU: the oracle at the geometry of BASELINE configs[3] / configs[4] against golden vectors produced by running the reference's
own head + criterion at that geometry (tests/golden/gen_golden_cfg34.py, round 6): PVTv2-B5 feature widths, 10-frame clips
(NUM_FRAMES = 10), K = 2 with ground truth on every frame (MS3) and K = 71 with the AVSS flag tensors at 512 x 512 (S = 5376
encoder tokens, 128 x 128 mask features).  Until round 5 the oracle was pinned at R50 widths / BT = 5 / K = 2 only."""
import json
import os

import numpy as np
import pytest
import torch

import gen_inputs
import synth
from oracle import combo_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
LEVELS = {"ms3_t10": ((7, 7), (14, 14), (28, 28)), "avss_512": ((16, 16), (32, 32), (64, 64))}


def load_case(case):
    z = np.load(os.path.join(G, f"head_{case}.npz"), allow_pickle=False)
    zc = np.load(os.path.join(G, f"criterion_{case}.npz"), allow_pickle=False)
    return z, zc, json.loads(str(z["case"]))


def case_inputs(case, c):
    feats, audio = gen_inputs.head_inputs(bt=c["bt"], hw=c["hw"], channels=tuple(c["channels"]), tag=f"feat.{case}")
    t_all = (gen_inputs.make_targets("all", bt=c["bt"], size=c["size"]) if c["K"] == 2
             else gen_inputs.make_targets_k(c["bt"], c["size"], c["K"], case))
    if c["crit"] == "ss":
        gt_index = torch.where(torch.tensor(c["gt_flag"]) == 1)[0]
        return feats, audio, [t_all[int(i)] for i in gt_index], gt_index
    return feats, audio, t_all, None


def frozen_choices(zc, n_over=37632):
    topk = torch.from_numpy(np.unpackbits(zc["topk_bits"], axis=2)[:, :, :n_over].astype(bool))
    return {"match_src": torch.from_numpy(zc["match_all_src"]), "match_tgt": torch.from_numpy(zc["match_all_tgt"]), "topk": topk}


def check_heads(case, z, logits, masks):
    """class logits and mask logits of all 10 prediction heads: the north-star's bound, 1e-3 * RMS(head) + 1e-3 * |ref|, on the
    8 192 sampled entries per head, no outlier budget"""
    worst = 0.0
    for i, (lg, m) in enumerate(zip(logits, masks)):
        for nm, t in ((f"dec/pred_logits{i}", lg), (f"dec/pred_masks{i}", m)):
            d = synth.unpack(nm, z)
            rms = float(d["l2"]) / np.sqrt(float(d["numel"]))
            worst = max(worst, synth.check_digest(t.detach().cpu(), d, nm, rtol=1e-3, atol=1e-3 * rms, k=8192, frac_bad=0.0) / rms)
    print(f"[{case}] all 10 heads within 1e-3 RMS + 1e-3 |ref|; worst sampled error {worst:.2e} RMS")


@pytest.mark.parametrize("case", ["ms3_t10", "avss_512"])
def test_oracle_head_and_criterion_at_pvt_geometry(case):
    z, zc, c = load_case(case)
    spec = json.loads(str(z["spec"]))
    P = synth.synth_state_dict(spec, 0)
    feats, audio, targets, gt_index = case_inputs(case, c)
    assert [t["labels"].tolist() for t in targets] == json.loads(str(zc["labels"]))
    masks_ref = synth.frozen_attn_masks(z, bt=c["bt"], sizes=LEVELS[case])
    with torch.no_grad():
        # (1) the oracle's own discrete choices: intermediates, the first heads, and the thresholded masks themselves
        out = O.head_forward(P, "", feats, audio, return_intermediates=True)
        it = out["_inter"]
        synth.check_digest(it["mask_features"], synth.unpack("pd/mask_features", z), "pd/mask_features", 1e-4, 2e-5)
        for i, m in enumerate(it["multi_scale"]):
            synth.check_digest(m, synth.unpack(f"pd/ms{i}", z), f"pd/ms{i}", 1e-4, 2e-5)
        synth.check_digest(it["fused_visual"], synth.unpack("fuse/visual", z), "fuse/visual", 1e-4, 2e-5)
        np.testing.assert_allclose(it["fused_audio"].numpy(), z["fuse/audio"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(it["audio256"].numpy(), z["fuse/audio256"], rtol=1e-4, atol=2e-5)
        assert tuple(out["middles_attn_mask"][0].shape) == tuple(z["dec/middle_shape"])
        cnt = np.array([int(a.sum()) for a in out["attn_masks"]])
        ref_cnt = z["dec/attn_true_count"]
        flips = [int((out["attn_masks"][i][::8] != masks_ref[i]).sum()) for i in range(9)]
        print(f"[{case}] attention-mask cells that differ from the reference's, heads 0..8: {flips} of {[m.numel() for m in masks_ref]}")
        assert np.abs(cnt - ref_cnt).max() <= 8 * 64, (cnt, ref_cnt)
        assert flips[0] <= 2  # head 0 sees no earlier mask: only round-off at the threshold can differ
        # (2) the reference's masks injected: every later head is the same continuous function
        out = O.head_forward(P, "", feats, audio, attn_override=masks_ref)
        logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
        masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
        assert tuple(torch.stack(logits).shape) == tuple(z["dec/logits_shape"])
        check_heads(case, z, logits, masks)
        # (3) the 39 losses with the reference's Hungarian pairs / point sets (criterion.py:233-287, criterion_ss.py:238-289;
        # the cosine loss groups the frames in fives whatever NUM_FRAMES says: criterion.py:284)
        torch.manual_seed(11)
        losses = O.set_criterion(out, targets, c["K"], gt_frame_index=gt_index, frozen=frozen_choices(zc))
        keys = json.loads(str(zc["keys"]))
        assert sorted(losses.keys()) == keys and len(keys) == 39
        got = np.array([float(losses[k]) for k in keys])
        np.testing.assert_allclose(got, zc["values"], rtol=5e-4, atol=2e-5)
        wd = O.loss_weights()
        np.testing.assert_allclose(float(sum(losses[k] * wd[k] for k in keys)), float(zc["total"]), rtol=5e-4)
        # (4) un-frozen: the oracle's own matcher finds the reference's pairs (scipy LSAP on the same costs)
        torch.manual_seed(11)
        sel = gt_index if gt_index is not None else torch.arange(c["bt"])
        idx = O.hungarian_matcher(out["pred_logits"][sel], out["pred_masks"][sel], targets)
        assert np.array_equal(np.concatenate([i.numpy() for i, _ in idx]), zc["match_all_src"][0])
        assert np.array_equal(np.concatenate([j.numpy() for _, j in idx]), zc["match_all_tgt"][0])


def test_oracle_gradients_at_ms3_t10_geometry():
    """gradient digests of 17 parameters + the head's inputs, every discrete choice frozen (224 x 224 case; the 512 x 512 backward
    of the CPU oracle is left to the GPU box's host cores: tests/test_head_cfg34_gpu.py)"""
    case = "ms3_t10"
    z, zc, c = load_case(case)
    P = synth.synth_state_dict(json.loads(str(z["spec"])), 0)
    grad_params = json.loads(str(zc["grad_params"]))
    feats, audio, targets, gt_index = case_inputs(case, c)
    for v in list(feats.values()) + [audio] + [P[p] for p in grad_params]:
        v.requires_grad_(True)
    out = O.head_forward(P, "", feats, audio, attn_override=synth.frozen_attn_masks(z, bt=c["bt"], sizes=LEVELS[case]))
    torch.manual_seed(11)
    losses = O.set_criterion(out, targets, c["K"], gt_frame_index=gt_index, frozen=frozen_choices(zc))
    wd = O.loss_weights()
    total = sum(losses[k] * wd[k] for k in losses)
    gi = list(feats.values()) + [audio] + [P[p] for p in grad_params]
    grads = torch.autograd.grad(total, gi, allow_unused=True)
    names = [f"feat.{k}" for k in feats] + ["feat.audio"] + grad_params
    for n, g in zip(names, grads):
        d = synth.unpack(f"grad/{n}", zc)
        scale = float(d["l2"]) / max(np.sqrt(float(d["numel"])), 1.0)
        if synth.upstream_of_sampling(n):  # (a bilinear tap within round-off of a pixel boundary: energy form, synth.py)
            synth.check_digest_l2(g, d, f"{case}/grad/{n}", rel_l2=1e-2, cap_rms=0.3, k=2048)
        else:
            # (0.5 %: twice the frames of head.npz - twice the ReLU cells of the FPN output convolution within round-off of 0, each
            #  moving a 3 x 3 x 64 patch of d loss / d res2; measured 0.39 % on feat.res2, <= 0.1 % elsewhere)
            synth.check_digest(g, d, f"{case}/grad/{n}", rtol=2e-3, atol=2e-3 * scale + 1e-9, k=2048, frac_bad=0.005)


def test_oracle_pvtv2_b5_matches_the_reference_golden():
    """oracle.pvtv2_b5 (functional restatement of backbone/pvtv2.py:60-175, 343-409) against tests/golden/pvt.npz - features and
    eight parameter gradients of the reference's own PyramidVisionTransformerV2 on name-seeded weights (gen_golden_pvt.py).  This
    pins the backbone half of oracle.maskformer_forward(backbone="pvt")."""
    z = np.load(os.path.join(G, "pvt.npz"), allow_pickle=False)
    spec = [(n, tuple(int(v) for v in s.split(","))) for n, s in zip(z["spec_names"].tolist(), z["spec_shapes"].tolist())]
    P = synth.synth_state_dict(spec, seed=0)
    probe = z["probe"].tolist()
    for p in probe:
        P[p].requires_grad_(True)
    x = synth.synth_tensor("pvt.x", (2, 3, 64, 64), 0)
    out = O.pvtv2_b5(P, "", x)
    names = ["res2", "res3", "res4", "res5"]
    assert [",".join(map(str, out[n].shape)) for n in names] == z["out_shapes"].tolist()
    for n in names:
        synth.check_digest(out[n], synth.unpack(f"out.{n}", z), f"pvt.out.{n}", rtol=2e-4, atol=2e-4)
    loss = sum((out[n] * synth.synth_tensor(f"pvt.g.{n}", tuple(out[n].shape), 0)).sum() for n in names)
    grads = torch.autograd.grad(loss, [P[p] for p in probe])
    for p, g in zip(probe, grads):
        d = synth.unpack(f"grad.{p}", z)
        synth.check_digest(g, d, f"pvt.grad.{p}", rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(d["sample"]).max())))
