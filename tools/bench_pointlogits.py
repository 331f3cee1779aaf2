#!/usr/bin/env python3
"""ns1: what a "fused mask_embed @ pixel_embed + loss" kernel would have to beat (bs = 8 x 5 frames, 10 prediction heads, Q = 100).

The matcher (matcher.py:92-131 of the reference) and the point losses (criterion.py:137-186) read the mask logits at 12 544
random points per frame - 4 x the 3 136 pixels of the 56 x 56 logit map.  Two ways to get those values:
  (a) as the step does: the full-resolution logits of all heads once (ONE exact-fp32 GEMM launch, 501 MB written) and bilinear
      gathers from them (csrc/matcher.hip, csrc/maskloss.hip);
  (b) the "fused" way: sample the pixel embedding at the points first (linear, so it commutes) and contract
      mask_embed [Q, 256] with the sampled embedding [256, 12 544] in a GEMM whose epilogue accumulates the cost / loss sums -
      the logits never reach HBM, but the contraction has 4 x the columns, per head.
Timed here: the contraction of (b) WITHOUT any epilogue work (exact fp32 and the 3-product split) against the whole of (a): the
logit GEMM + the matcher's gather kernel.  The cosine term (criterion.py:208-231) needs every pixel of every query of the 9
intermediate heads in either design; its kernels run at 5.1 / 5.5 TB/s (cosine_stats 89 us for 451 MB, cosine_grad 164 us for
902 MB in profiles/r04_*steady_state*): fusing the statistics into the GEMM epilogue would save the 89 us re-read."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import combo_avs_amd  # noqa: E402,F401
from combo_avs_amd.ops import masklogit  # noqa: E402

HEADS, BT, Q, C, HW, P = 10, 40, 100, 256, 56 * 56, 12544


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


torch.manual_seed(0)
me = [torch.randn(BT, Q, C, device="cuda") for _ in range(HEADS)]
mf = torch.randn(BT, HW, C, device="cuda")
mf_pts = torch.randn(BT, P, C, device="cuda")  # the pixel embedding sampled at the matcher's points (values do not matter here)
out = torch.empty(HEADS, BT, Q, HW, device="cuda")
out_pts = torch.empty(HEADS, BT, Q, P, device="cuda")
t_full = timeit(lambda: masklogit.mask_logits_all_into(me, mf, out))
t_pts = timeit(lambda: masklogit.mask_logits_all_into(me, mf_pts, out_pts))
img = masklogit.presplit_batched(mf_pts, transpose=False)
me_c = [m.contiguous() for m in me]


def x3():
    for h in range(HEADS):
        masklogit.gemm_nt_batched(me_c[h], img, out_pts[h])


t_pts_x3 = timeit(x3)
gf = 2.0 * HEADS * BT * Q * C / 1e9
print(f"(a) full-resolution logits, all heads, exact fp32: {t_full:8.1f} us ({gf * HW / t_full * 1e3:6.1f} TFLOP/s, {HEADS * BT * Q * HW * 4 / 1e6:.0f} MB written)")
print(f"(b) logits at the {P} points, exact fp32:        {t_pts:8.1f} us ({gf * P / t_pts * 1e3:6.1f} TFLOP/s) - before any epilogue work")
print(f"(b) logits at the {P} points, 3-product split:   {t_pts_x3:8.1f} us ({gf * P / t_pts_x3 * 1e3:6.1f} TFLOP/s) - before any epilogue work")
print("    in the step (profiles/r04_*steady_state*): logit GEMM 1 249 us + matcher_cost_kernel 284 us + uncertain_select 226 us")
