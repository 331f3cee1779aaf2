#!/usr/bin/env python3
"""Where does the fp16-piece forward mode run out of RANGE?  The 40-step x 20-lr loop of tests/test_graph_gpu.py::
test_graphed_training_actually_learns, eager, with the largest |operand| of every 3-product forward GEMM / convolution recorded per
call site (shape), and the first step at which a loss is not finite.
    python tools/probe_f16_range.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import graph_compare as GC  # noqa: E402
from combo_avs_amd.ops import linear as L  # noqa: E402
from combo_avs_amd.ops import conv3x3 as C3  # noqa: E402
from combo_avs_amd.trainer import train_step  # noqa: E402

model, opt, batches, state = GC.build("r50")
for s in opt.segments:
    s[2] = s[2] * 20.0
seen = {}
own = L.gemm_nt_bf16


def spy(a, w, bias=None, relu=False, out=None, img=None):
    k = ("gemm", tuple(a.shape), tuple(w.shape))
    seen[k] = max(seen.get(k, (0.0, 0.0))[0], float(a.abs().max())), max(seen.get(k, (0.0, 0.0))[1], float(w.abs().max()))
    return own(a, w, bias, relu, out, img)


L.gemm_nt_bf16 = spy
own_conv = C3._conv_tokens


def spy_conv(x_tok, wm, bias, B, H, W, cin, cout, exact, relu=False):
    k = ("conv3x3", tuple(x_tok.shape), tuple(wm.shape))
    seen[k] = max(seen.get(k, (0.0, 0.0))[0], float(x_tok.abs().max())), max(seen.get(k, (0.0, 0.0))[1], float(wm.abs().max()))
    return own_conv(x_tok, wm, bias, B, H, W, cin, cout, exact, relu)


C3._conv_tokens = spy_conv
for step in range(40):
    losses = train_step(model, opt, batches[0])
    tot = float(sum(losses.values()))
    big = {k: v for k, v in seen.items() if v[0] > 1000 or v[1] > 100}
    print(f"step {step}: total {tot:.3f}; operands beyond |a| 1000 / |w| 100: {big}", flush=True)
    if tot != tot:
        break
top = sorted(seen.items(), key=lambda kv: -kv[1][0])[:8]
print("largest activations:", [(k, f"{v[0]:.3g}", f"{v[1]:.3g}") for k, v in top])
