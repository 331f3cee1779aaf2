#!/usr/bin/env python3
"""End-to-end effect of the backbones' 3-product forward (ops.convwrw.FWD_X3) on the mask logits of all 10 prediction heads:
the share of logits beyond the north-star's bound (1e-3 RMS(head) + 1e-3 |ref|) between a forward with the own 3-product
convolutions and one with the library's fp32 convolutions - next to the same share between TWO library forwards (their stride-2 /
VGGish kernels accumulate with atomics: the chaos floor of this comparison, a flipped attention-mask cell re-routes a query)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import graph_compare as GC  # noqa: E402

from combo_avs_amd.ops import convwrw  # noqa: E402

model, opt, batches, _ = GC.build("r50")
dec = model.sem_seg_head.predictor
grabbed = {}
dec.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o["_logits_all"].detach().clone()))


def run(x3):
    convwrw.FWD_X3 = x3
    model(batches[0])
    return grabbed["logits"]  # [heads, BT, Q, h, w]


def beyond(a, ref):
    out = []
    for h in range(ref.shape[0]):
        rms = ref[h].pow(2).mean().sqrt()
        out.append(float(((a[h] - ref[h]).abs() > 1e-3 * rms + 1e-3 * ref[h].abs()).float().mean()))
    return out


lib1, lib2, own1, own2 = run(False), run(False), run(True), run(True)
fmt = lambda v: " ".join(f"{x:.2e}" for x in v)  # noqa: E731
print("share of mask logits beyond 1e-3 RMS + 1e-3 |ref|, heads 0..9")
print("library vs library (two runs):", fmt(beyond(lib2, lib1)))
print("own x3     vs library        :", fmt(beyond(own1, lib1)))
print("own x3     vs own x3 (2 runs):", fmt(beyond(own2, own1)))
print("max |own - library| / RMS per head:", fmt([float((own1[h] - lib1[h]).abs().max() / lib1[h].pow(2).mean().sqrt()) for h in range(lib1.shape[0])]))
