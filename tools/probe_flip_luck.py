#!/usr/bin/env python3
"""How much of the head-level "0 outliers" result is the arithmetic and how much the fixture (round 6)?  The golden head fixture
(tests/golden/head.npz: the REFERENCE's fp32 CPU outputs, BT = 5) through the product head with the forward GEMMs of (a) the default
exact-fp32 kernels, (b) the LIBRARY's fp32 GEMMs (another legitimate fp32 summation order) in the pixel decoder's encoder, (c) the
fp16-piece 3-product mode in the pixel decoder's encoder, (d) ... in the whole head, (e) the bf16-piece 3-product mode in the whole head.
Per variant: attention-mask cells that differ from the reference's (of 1.6 M; the decoder re-routes a query whose cell flips), class
logits / sampled mask logits beyond the north-star bound (1e-3 x RMS + 1e-3 x |ref|), relative L2 of the mask features.
    python tools/probe_flip_luck.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import combo_avs_amd  # noqa: F401,E402
from combo_avs_amd.modeling import pixel_decoder as PD  # noqa: E402
from combo_avs_amd.ops import linear as L  # noqa: E402
import gen_inputs  # noqa: E402
import synth  # noqa: E402
from test_head_gpu import build_head  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "head.npz"))
spec = json.loads(str(z["spec"]))
head, cfg = build_head()
head.load_state_dict(synth.synth_state_dict(spec, 0))
head = head.cuda().eval()
ref_masks = synth.frozen_attn_masks(z)


def run(layout, enc="fp32", fpn="fp32", glob="fp32", library_encoder=False):
    feats, audio = gen_inputs.head_inputs()
    feats = {k: (v.cuda().contiguous(memory_format=torch.channels_last) if layout == "channels_last" else v.cuda()) for k, v in feats.items()}
    saved = (PD.PIXEL_DECODER_FORWARD, PD.PIXEL_DECODER_FPN_FORWARD, L.gemm_nt_f32)
    PD.PIXEL_DECODER_FORWARD, PD.PIXEL_DECODER_FPN_FORWARD = enc, fpn
    L.set_forward_precision(glob)
    if library_encoder:
        own = L.gemm_nt_f32

        def lib_gemm(a, w, bias=None, relu=False, out=None):
            if a.shape[0] < 20000:  # (the encoder's 26 460-token GEMMs only)
                return own(a, w, bias, relu, out)
            y = torch.nn.functional.linear(a, w, bias)
            y = torch.relu_(y) if relu else y
            if out is not None:
                out.copy_(y)
                return out
            return y
        L.gemm_nt_f32 = lib_gemm
    try:
        with torch.no_grad(), L.grouped_presplit():
            out = head(dict(feats), audio.cuda())
            mf, _, ms = head.pixel_decoder.forward_features(dict(feats))
        torch.cuda.synchronize()
    finally:
        PD.PIXEL_DECODER_FORWARD, PD.PIXEL_DECODER_FPN_FORWARD, L.gemm_nt_f32 = saved
        L.set_forward_precision(L.DEFAULT_FORWARD_PRECISION)
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    flips, beyond, n_s, worst = 0, 0, 0, 0.0
    for i, m in enumerate(masks):
        d = synth.unpack(f"dec/pred_masks{i}", z)
        idx = synth.digest_indices(m.numel(), 4096, f"dec/pred_masks{i}")
        got = m.reshape(-1).cpu().numpy()[idx].astype(np.float64)
        ref = np.asarray(d["sample"]).astype(np.float64)
        rms = float(np.sqrt((ref ** 2).mean()))
        err = np.abs(got - ref)
        beyond += int((err > 1e-3 * rms + 1e-3 * np.abs(ref)).sum())
        n_s += err.size
        worst = max(worst, float(err.max() / rms))
        if i < len(ref_masks):
            tgt = [(7, 7), (14, 14), (28, 28)][i % 3]
            down = torch.nn.functional.interpolate(m, size=tgt, mode="bilinear", align_corners=False)
            flips += int(((down.sigmoid().flatten(2) < 0.5).cpu() != ref_masks[i]).sum())
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]).cpu().numpy()
    ref = z["dec/pred_logits"]
    bad = int((np.abs(logits - ref) > 1e-3 * np.sqrt((ref ** 2).mean()) + 1e-3 * np.abs(ref)).sum())
    d = synth.unpack("pd/mask_features", z)
    idx = synth.digest_indices(mf.numel(), 4096, "pd/mask_features")
    g, r = mf.reshape(-1).cpu().numpy()[idx].astype(np.float64), np.asarray(d["sample"]).astype(np.float64)
    l2 = float(np.sqrt(((g - r) ** 2).sum() / (r ** 2).sum()))
    return flips, bad, logits.size, beyond, n_s, worst, l2


for layout in ("nchw (the golden tests' input layout: the pixel decoder's convolutions on the library)", "channels_last (the step's layout: own kernels)"):
    print(f"== input features {layout}")
    lay = layout.split()[0]
    for name, kw in [("default: exact fp32 kernels", {}),
                     ("library fp32 GEMMs in the pixel decoder's encoder", {"library_encoder": True}),
                     ("fp16 x3 in the pixel decoder's encoder", {"enc": "f16x3"}),
                     ("fp16 x3 in the whole pixel decoder", {"enc": "f16x3", "fpn": "f16x3"}),
                     ("fp16 x3 in the whole head", {"glob": "f16x3"}),
                     ("bf16 x3 in the whole pixel decoder", {"enc": "x3", "fpn": "x3"}),
                     ("bf16 x3 in the whole head", {"glob": "x3"})]:
        f, bad, nl, beyond, ns, worst, l2 = run(lay, **kw)
        print(f"{name:52s} flipped mask cells {f:4d} | class logits beyond {bad:4d} of {nl} | sampled mask logits beyond {beyond:5d} of {ns} "
              f"(worst {worst:.2e} RMS) | mask features rel L2 {l2:.2e}", flush=True)
