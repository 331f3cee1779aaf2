#!/usr/bin/env python3
"""Micro-benchmark of the input-gradient GEMM (csrc/gemm_nt3.hip: fp32 accuracy from 3 bf16 MFMA products) on the
dX shapes of the training step, against its two ceilings - 833 TFLOP/s useful (2.5 PF / 3 products) and the HBM time of the
algorithmic bytes (A read once, C written once, weight image once) - and against the library's fp32 and bf16 GEMMs.

    python tools/bench_nt3.py [--iters N] [--shapes all|small|big] [--no-lib]
    COMBO_NT3_DBG=<bits>  ablation instances: one of 1 2 4 8 16 32 41 63 (1 no DMA, 2 no LDS reads, 4 no barrier, 8 no stores, 16 no split, 32 no MFMA)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import combo_avs_amd  # noqa: F401,E402
from combo_avs_amd.ops.linear import gemm_nt_x3, presplit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--shapes", default="all")
ap.add_argument("--no-lib", action="store_true")
ap.add_argument("--tile", type=int, default=0, help="force the tile shape (combo_gemm_nt_x3_tile): 1 wide, 2 mid, 3 skinny, 4 tall")
args = ap.parse_args()

# (M, K, N) of dX = dY[M, K] . W^T-image[N, K]: K = the layer's output features, N = its input features
SHAPES = [(41160, 256, 256), (41160, 288, 256), (41160, 1024, 256), (41160, 256, 1024), (125440, 256, 256), (31360, 256, 512),
          (7840, 256, 1024), (1960, 256, 2048), (4000, 256, 256), (4000, 2048, 256), (4000, 256, 2048), (4000, 512, 256),
          (125440, 64, 256), (125440, 256, 64), (31360, 512, 128), (31360, 128, 512), (7840, 1024, 256), (7840, 256, 1024),
          (1960, 2048, 512), (1960, 512, 2048)]
if args.shapes == "small":
    SHAPES = [(41160, 256, 256), (41160, 1024, 256), (41160, 256, 1024)]
if args.shapes == "big":
    SHAPES = SHAPES[:5]
if args.shapes == "round":  # exactly one round of wide tiles (256 x 128) on 256 CUs: per-stage cost without quantisation effects
    SHAPES = [(32768, 256, 256), (32768, 1024, 256), (8192, 256, 1024), (32768, 2048, 256)]
if args.tile:
    from combo_avs_amd import _lib
    _lib.lib().combo_gemm_nt_x3_tile(args.tile)


def timeit(fn, n):
    t0 = time.time()
    while time.time() - t0 < 0.3:  # the clocks need ~0.3 s to settle (tools/clock_probe_f32.py)
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


torch.manual_seed(0)
for M, K, N in SHAPES:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    img = presplit(w)
    us = timeit(lambda: gemm_nt_x3(a, w, img=img), args.iters)
    flops = 2.0 * M * N * K
    byts = 4.0 * (M * K + N * K + M * N)
    line = (f"[x3 {M}x{K}->{N}] own {us:7.1f} us = {flops / us / 1e6:6.1f} TF/s useful ({flops / us / 1e6 / 833.3 * 100:4.1f} % of 833), "
            f"{byts / us / 1e3:6.0f} GB/s ({byts / us / 1e3 / 8000 * 100:4.1f} % of 8 TB/s); floors: mfma {flops / 833.3e6:5.1f} us, hbm@6.3TB/s {byts / 6.3e6:5.1f} us")
    if not args.no_lib:
        lib = timeit(lambda: torch.nn.functional.linear(a, w), args.iters)
        ab, wb = a.bfloat16(), w.bfloat16()
        libb = timeit(lambda: torch.nn.functional.linear(ab, wb), args.iters)
        line += f" | library fp32 {lib:7.1f} us, bf16 (1 product, bf16 in/out) {libb:6.1f} us"
    print(line, flush=True)
