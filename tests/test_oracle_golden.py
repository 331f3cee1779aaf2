"""CPU: the oracle (oracle/combo_oracle.py) against golden vectors produced by running the reference's own
code (tests/golden/gen_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from oracle import combo_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ---------------------------------------------------------------------------------------- a6 core op
@pytest.mark.parametrize("tag", ["t_double", "t_float", "t_grad30", "t_grad32", "t_grad64", "t_grad71", "t_grad1025", "edge"])
def test_msda_core_reference_unit_cases(tag):
    z = load("msda_core.npz")
    dt = torch.float32 if tag == "t_float" else torch.float64
    value = torch.from_numpy(z[f"{tag}/value"]).to(dt).requires_grad_(True)
    loc = torch.from_numpy(z[f"{tag}/loc"]).to(dt).requires_grad_(True)
    w = torch.from_numpy(z[f"{tag}/w"]).to(dt).requires_grad_(True)
    shapes = z[f"{tag}/shapes"].tolist()
    out = O.ms_deform_attn_core(value, shapes, loc, w)
    tol = dict(rtol=1e-5, atol=1e-8) if dt == torch.float32 else dict(rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out.detach().numpy(), z[f"{tag}/out"], **tol)
    gv, gl, gw = torch.autograd.grad(out, (value, loc, w), torch.from_numpy(z[f"{tag}/grad_out"]).to(dt))
    np.testing.assert_allclose(gv.numpy(), z[f"{tag}/grad_value"], **tol)
    np.testing.assert_allclose(gw.numpy(), z[f"{tag}/grad_w"], **tol)
    # d/dloc is discontinuous exactly on pixel borders; the edge case sits on them on purpose
    if tag != "edge":
        np.testing.assert_allclose(gl.numpy(), z[f"{tag}/grad_loc"], **tol)


def prod_inputs():
    shapes = [(7, 7), (14, 14), (28, 28)]
    B, S = 2, 1029
    v = synth.synth_tensor("prod.value", (B, S, 8, 32), 0)
    refp = synth.synth_tensor("prod.ref", (B, S, 1, 1, 1, 2), 0, kind="unit")
    off = synth.synth_tensor("prod.off", (B, S, 8, 3, 4, 2), 0, scale=2.5)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = refp + off / norm[None, None, None, :, None, :]
    w = torch.softmax(synth.synth_tensor("prod.w", (B, S, 8, 12), 0), -1).view(B, S, 8, 3, 4)
    return v, shapes, loc, w


def test_msda_core_production_shape():
    z = load("msda_core.npz")
    v, shapes, loc, w = prod_inputs()
    v.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    out = O.ms_deform_attn_core(v, shapes, loc, w)
    go = synth.synth_tensor("prod.grad_out", tuple(out.shape), 0)
    gv, gl, gw = torch.autograd.grad(out, (v, loc, w), go)
    for nm, t in (("out", out), ("grad_value", gv), ("grad_loc", gl), ("grad_w", gw)):
        synth.check_digest(t, synth.unpack(f"prod/{nm}", z), f"prod/{nm}", rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------------------------------- a3, a1
def test_position_embedding_sine():
    z = load("pe_sine.npz")
    np.testing.assert_allclose(O.position_embedding_sine(1, 7, 7).numpy(), z["pe7/full"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.position_embedding_sine(2, 5, 9).numpy(), z["pe5x9/full"], rtol=1e-6, atol=1e-6)
    for hw in (14, 28, 56):
        synth.check_digest(O.position_embedding_sine(1, hw, hw), synth.unpack(f"pe{hw}", z), f"pe{hw}", 1e-6, 1e-6)


@pytest.mark.parametrize("dim", [256, 512])
def test_sem_mix(dim):
    z = load("sem_mix.npz")
    spec = json.loads(str(z[f"sem{dim}/spec"]))
    P = {"m.0." + k: synth.synth_param(f"sem{dim}." + k, s) for k, s in spec}
    f = synth.synth_tensor(f"sem{dim}.f", (3, dim, 6, 5), 0)
    p = synth.synth_tensor(f"sem{dim}.p", (3, dim, 6, 5), 0)
    np.testing.assert_allclose(O.channel_weighted_gate(P, "m.0.", p).numpy(), z[f"sem{dim}/gate"], rtol=1e-5, atol=1e-6)
    mixed = O.sem_mix(P, "m.", {"res2": f}, {"res2": p})["res2"]
    np.testing.assert_allclose(mixed.numpy(), z[f"sem{dim}/mixed"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------- head (a2,a4,a5,a7-a13)
@pytest.fixture(scope="module")
def head_run():
    import gen_inputs
    z = load("head.npz")
    spec = json.loads(str(z["spec"]))
    P = synth.synth_state_dict(spec, 0)
    feats, audio = gen_inputs.head_inputs()
    for v in feats.values():
        v.requires_grad_(True)
    audio.requires_grad_(True)
    out = O.head_forward(P, "", feats, audio, return_intermediates=True)
    return z, P, feats, audio, out


def test_head_pixel_decoder(head_run):
    z, P, feats, audio, out = head_run
    it = out["_inter"]
    synth.check_digest(it["mask_features"], synth.unpack("pd/mask_features", z), "pd/mask_features", 1e-4, 2e-5)
    for i, m in enumerate(it["multi_scale"]):
        synth.check_digest(m, synth.unpack(f"pd/ms{i}", z), f"pd/ms{i}", 1e-4, 2e-5)


def test_head_avfuse_and_audio_mlp(head_run):
    z, P, feats, audio, out = head_run
    it = out["_inter"]
    synth.check_digest(it["fused_visual"], synth.unpack("fuse/visual", z), "fuse/visual", 1e-4, 2e-5)
    np.testing.assert_allclose(it["fused_audio"].detach().numpy(), z["fuse/audio"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(it["audio256"].detach().numpy(), z["fuse/audio256"], rtol=1e-4, atol=2e-5)


def test_head_decoder_outputs(head_run):
    z, P, feats, audio, out = head_run
    logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    np.testing.assert_allclose(torch.stack(logits).detach().numpy(), z["dec/pred_logits"], rtol=1e-3, atol=1e-4)
    for i, m in enumerate(masks):
        # BASELINE.json tolerance: 1e-3 rel on mask logits (scale-relative: logits have std ~6)
        synth.check_digest(m, synth.unpack(f"dec/pred_masks{i}", z), f"dec/pred_masks{i}", rtol=1e-3, atol=1e-3)
    cnt = np.array([int(a.sum()) for a in out["attn_masks"]])
    assert np.abs(cnt - z["dec/attn_true_count"]).max() <= 8, (cnt, z["dec/attn_true_count"])
    used = np.array([int(a.sum()) for a in out["attn_masks_used"]] + [cnt[-1]])  # last head's mask is never used
    assert np.abs(used - z["dec/attn_used_true_count"]).max() <= 8, (used, z["dec/attn_used_true_count"])
    assert (used[:3] < cnt[:3]).all()  # the fully-blocked-row reset (transformer_decoder.py:458) is exercised
    bits = np.packbits(out["attn_masks"][0][0].numpy().astype(np.uint8))
    assert (np.unpackbits(bits) != np.unpackbits(z["dec/attn0_bits"])).sum() <= 1
    assert len(out["middles_attn_mask"]) == 9


def test_audio_scramble_rule():
    BT, Q = 5, 100
    a = torch.arange(BT, dtype=torch.float32).view(BT, 1, 1).repeat(1, 1, 4)
    s = O.scramble_audio(a, Q)
    for q in (0, 19, 20, 57, 99):
        for b in range(BT):
            assert s[q, b, 0].item() == (q * BT + b) // Q


def test_inference_tail(head_run):
    z, P, feats, audio, out = head_run
    zi = load("inference.npz")
    with torch.no_grad():
        sem = O.semantic_inference(out["pred_logits"], out["pred_masks"], (224, 224))
    synth.check_digest(sem, synth.unpack("sem_seg", zi), "sem_seg", rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(sem[0, :, ::8, ::8].numpy(), zi["sem_seg_frame0_ds"], rtol=1e-3, atol=1e-3)


# ---------------------------------------------------------------------------------------- criterion (a14-a16)
@pytest.mark.parametrize("mode", ["s4", "all", "ss"])
def test_criterion_losses_and_grads(head_run, mode):
    import gen_inputs
    z, P, feats, audio, out = head_run
    zc = load("criterion.npz")
    torch.manual_seed(11)
    if mode == "ss":
        gt_flag = torch.from_numpy(zc["ss/gt_flag"])
        t_all = gen_inputs.make_targets("all")
        targets = [t_all[i] for i in range(5) if gt_flag[i] == 1]
        losses = O.set_criterion(out, targets, 2, gt_frame_index=torch.where(gt_flag == 1)[0])
    else:
        targets = gen_inputs.make_targets(mode)
        losses = O.set_criterion(out, targets, 2)
    keys = json.loads(str(zc[f"{mode}/keys"]))
    assert sorted(losses.keys()) == keys and len(keys) == 39
    got = np.array([float(losses[k]) for k in keys])
    np.testing.assert_allclose(got, zc[f"{mode}/values"], rtol=2e-3, atol=1e-4)
    wd = O.loss_weights()
    total = sum(losses[k] * wd[k] for k in keys)
    np.testing.assert_allclose(float(total), float(zc[f"{mode}/total"]), rtol=1e-3)
    # matcher indices of the final layer under the same RNG state
    torch.manual_seed(11)
    sel = {"s4": torch.arange(0, 5, 5), "all": torch.arange(5), "ss": torch.where(torch.from_numpy(zc["ss/gt_flag"]) == 1)[0]}[mode]
    idx = O.hungarian_matcher(out["pred_logits"][sel].detach(), out["pred_masks"][sel].detach(), targets)
    assert np.array_equal(np.stack([i.numpy() for i, _ in idx]), zc[f"{mode}/match_src"])
    assert np.array_equal(np.stack([j.numpy() for _, j in idx]), zc[f"{mode}/match_tgt"])
    grad_params = json.loads(str(zc["grad_params"]))
    for p in grad_params:
        P[p].requires_grad_(True)
    # P entries were not requiring grad during the forward above -> recompute the forward with grads on
    out2 = O.head_forward(P, "", feats, audio)
    torch.manual_seed(11)
    losses2 = (O.set_criterion(out2, targets, 2, gt_frame_index=torch.where(gt_flag == 1)[0]) if mode == "ss"
               else O.set_criterion(out2, targets, 2))
    total2 = sum(losses2[k] * wd[k] for k in keys)
    gi = list(feats.values()) + [audio] + [P[p] for p in grad_params]
    grads = torch.autograd.grad(total2, gi, allow_unused=True)
    names = [f"feat.{k}" for k in feats] + ["feat.audio"] + grad_params
    for n, g in zip(names, grads):
        d = synth.unpack(f"{mode}/grad/{n}", zc)
        scale = float(d["l2"]) / max(np.sqrt(float(d["numel"])), 1.0)
        # Noise floor of this comparison: the oracle differs from the reference only by fp32 re-association, yet in the modes
        # with ground truth on every frame up to 1.6 % of the sampled gradient entries of the 7x7-level parameters move by more
        # than 5e-3 of the tensor's RMS (s4: none) - a near-zero attention-mask cell or a top-k tie of the importance
        # sampling that falls the other way changes one query's gradient wholesale.  The GPU tests inherit this floor.
        # The 3 % allowance applies ONLY to the tensor where that noise was measured (input_proj.0.0.weight: 0.56 % / 1.56 %);
        # everything else stays at 1 % (measured <= 0.46 %).  The frozen-choices test below removes the noise altogether.
        noisy = mode != "s4" and n == "pixel_decoder.input_proj.0.0.weight"
        synth.check_digest(g, d, f"{mode}/grad/{n}", rtol=5e-3, atol=5e-3 * scale + 1e-9, frac_bad=0.03 if noisy else 0.01)
    for p in grad_params:
        P[p].requires_grad_(False)


PIXEL_BOUNDARY = synth.PIXEL_BOUNDARY  # (tests/golden/synth.py)


@pytest.mark.parametrize("group", ["elementwise_2e-3", "pixel_boundary_tensors_relative_L2_1e-2_only"])
@pytest.mark.parametrize("mode", ["s4", "all", "ss"])
def test_criterion_grads_with_the_references_discrete_choices_frozen(head_run, mode, group):
    """The same gradient comparison with the reference's own discrete choices injected - the 9 attention masks
    (`dec/attn_bits*`), the Hungarian pairs of all 10 outputs (`*/match_all_*`) and the top-k sets of the importance sampling
    (`*/topk_bits`).  What remains is floating-point re-association: NO outlier budget at 2e-3 for the gradients that do not
    pass through the deformable encoder's bilinear sampling, and an energy bound (synth.check_digest_l2: relative L2 error
    <= 1e-2, no entry beyond 0.3 RMS) for the four that do - a tap within round-off of a pixel boundary is the one discrete
    event that cannot be injected (measured with everything else frozen: the 8.6 % / 15.5 % "outliers" of
    input_proj.0.0.weight are 2-3 such taps, each moving thousands of entries by ~1e-3 of the RMS; relative L2 error 1.9e-3)."""
    import gen_inputs
    z, P, feats, audio, out = head_run
    zc = load("criterion.npz")
    grad_params = json.loads(str(zc["grad_params"]))
    for p in grad_params:
        P[p].requires_grad_(True)
    try:
        out2 = O.head_forward(P, "", feats, audio, attn_override=synth.frozen_attn_masks(z))
        fr = synth.frozen_criterion_choices(zc, mode)
        torch.manual_seed(11)
        if mode == "ss":
            gt_flag = torch.from_numpy(zc["ss/gt_flag"])
            t_all = gen_inputs.make_targets("all")
            targets = [t_all[i] for i in range(5) if gt_flag[i] == 1]
            losses = O.set_criterion(out2, targets, 2, gt_frame_index=torch.where(gt_flag == 1)[0], frozen=fr)
        else:
            targets = gen_inputs.make_targets(mode)
            losses = O.set_criterion(out2, targets, 2, frozen=fr)
        keys = json.loads(str(zc[f"{mode}/keys"]))
        got = np.array([float(losses[k]) for k in keys])
        np.testing.assert_allclose(got, zc[f"{mode}/values"], rtol=5e-4, atol=2e-5)
        wd = O.loss_weights()
        total = sum(losses[k] * wd[k] for k in keys)
        gi = list(feats.values()) + [audio] + [P[p] for p in grad_params]
        grads = torch.autograd.grad(total, gi, allow_unused=True)
        names = [f"feat.{k}" for k in feats] + ["feat.audio"] + grad_params
        for n, g in zip(names, grads):
            d = synth.unpack(f"{mode}/grad/{n}", zc)
            scale = float(d["l2"]) / max(np.sqrt(float(d["numel"])), 1.0)
            # measured (oracle vs reference): 0.000 % outliers for every tensor that is not downstream of the deformable
            # encoder's sampling; relative L2 error <= 2.7e-3 and worst entry <= 0.11 RMS for those that are (PIXEL_BOUNDARY)
            if (n in PIXEL_BOUNDARY) != group.startswith("pixel_boundary"):
                continue
            if n in PIXEL_BOUNDARY:  # CPU oracle vs CPU reference: the same pixel-boundary effect, hence the same looser class
                synth.check_digest_l2(g, d, f"{mode}/grad/{n}", rel_l2=1e-2, cap_rms=0.3)
            else:
                synth.check_digest(g, d, f"{mode}/grad/{n}", rtol=2e-3, atol=2e-3 * scale + 1e-9, frac_bad=0.002)
    finally:
        for p in grad_params:
            P[p].requires_grad_(False)


# ---------------------------------------------------------------------------------------- plain-C core-op oracle
@pytest.mark.parametrize("tag", ["t_double", "t_float", "t_grad30", "t_grad32", "t_grad71", "edge"])
def test_c_oracle_reference_unit_cases(tag):
    from oracle import msda_c
    z = load("msda_core.npz")
    dt = np.float32 if tag == "t_float" else np.float64
    value, loc, w = (z[f"{tag}/{k}"].astype(dt) for k in ("value", "loc", "w"))
    shapes = z[f"{tag}/shapes"]
    tol = dict(rtol=1e-5, atol=1e-8) if dt == np.float32 else dict(rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(msda_c.forward(value, shapes, loc, w), z[f"{tag}/out"], **tol)
    gv, gl, gw = msda_c.backward(z[f"{tag}/grad_out"].astype(dt), value, shapes, loc, w)
    np.testing.assert_allclose(gv, z[f"{tag}/grad_value"], **tol)
    np.testing.assert_allclose(gw, z[f"{tag}/grad_w"], **tol)
    if tag != "edge":
        np.testing.assert_allclose(gl, z[f"{tag}/grad_loc"], **tol)


def test_c_oracle_equals_torch_oracle_production_shape():
    from oracle import msda_c
    v, shapes, loc, w = prod_inputs()
    out = O.ms_deform_attn_core(v, shapes, loc, w)
    np.testing.assert_allclose(msda_c.forward(v.numpy(), shapes, loc.numpy(), w.numpy()), out.numpy(), rtol=1e-5, atol=1e-5)
