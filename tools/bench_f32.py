#!/usr/bin/env python3
"""Micro-benchmark of csrc/gemm_f32.hip (exact-fp32 MFMA GEMM) on the head's forward shapes, against the fp32 MFMA peak
(157.3 TFLOP/s) and the library's fp32 GEMM.  `--iters N` back-to-back launches per shape after a warm-up that lets the clocks
settle; `--shapes small` limits the run for PMC passes (tools/pmc_f32.sh).  COMBO_F32_TILE=1/2/3 forces wide/mid/skinny."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import combo_avs_amd  # noqa: F401,E402
from combo_avs_amd.ops.linear import gemm_nt_f32  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--shapes", default="all")
ap.add_argument("--no-lib", action="store_true")
ap.add_argument("--no-splitk", action="store_true")
args = ap.parse_args()

SHAPES = [(41160, 256, 1024), (41160, 1024, 256), (41160, 256, 256), (41160, 256, 288), (125440, 256, 256), (31360, 256, 256),
          (4000, 256, 256), (4000, 256, 2048), (4000, 2048, 256), (4000, 256, 512), (7840, 256, 256), (1960, 256, 256),
          (16384, 2048, 256), (7840, 1024, 256), (1960, 2048, 256), (41160, 256, 96)]
if args.shapes == "small":
    SHAPES = SHAPES[:2]
if args.shapes == "fixed":  # one round of wide tiles at growing K: the intercept is the per-launch fixed cost
    SHAPES = [(65536, 16, 128), (65536, 64, 128), (65536, 256, 128), (65536, 512, 128), (65536, 1024, 128), (65536, 2048, 128),
              (131072, 256, 128), (131072, 1024, 128)]


def timeit(fn, n):
    # ~0.3 s of back-to-back launches first: the clocks need that long to settle (a cold 50-launch sample under-reports this
    # kernel by ~15 %: 222 vs 193 us at 41160 x 256 -> 1024, tools/clock_probe_f32.py)
    import time
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


if args.no_splitk:
    import combo_avs_amd.ops.linear as _L
    _L.SPLITK_F32 = False
torch.manual_seed(0)
for M, K, N in SHAPES:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    us = timeit(lambda: gemm_nt_f32(a, w, b, True, out=out), args.iters)
    line = f"[f32 {M}x{K}->{N}] own {us:7.1f} us = {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s ({2.0 * M * N * K / us / 1e6 / 157.3 * 100:4.1f} % of 157.3)"
    if not args.no_lib:
        lib = timeit(lambda: torch.relu_(torch.nn.functional.linear(a, w, b)), args.iters)
        line += f" | library fp32 (+relu) {lib:7.1f} us"
    print(line, flush=True)
