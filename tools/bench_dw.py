#!/usr/bin/env python3
"""dW = dY^T X (+ db) for the decoder / pixel-decoder shapes: csrc/gemm_tn.hip (+ fused reduce) vs the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd.ops import linear as L


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


shapes = [(4000, 256, 256), (4000, 2048, 256), (4000, 256, 2048), (4000, 3, 256), (1960, 256, 256), (7840, 256, 256),
          (31360, 256, 256), (41160, 256, 256), (41160, 1024, 256), (41160, 256, 1024), (41160, 192, 256), (125440, 256, 256)]
for M, N, K in shapes:
    dy = torch.randn(M, N, device="cuda")
    x = torch.randn(M, K, device="cuda")
    ref = dy.double().t() @ x.double()

    def lib():
        torch.backends.cuda.matmul.allow_tf32 = False
        return dy.t() @ x, dy.sum(0)

    def lib3():
        torch.backends.cuda.matmul.allow_tf32 = True
        r = dy.t() @ x, dy.sum(0)
        torch.backends.cuda.matmul.allow_tf32 = False
        return r

    def mine():
        return L.gemm_tn_x3(dy, x, with_bias_grad=True)

    t1, t2, t3 = timeit(lib), timeit(lib3), timeit(mine)
    err = float(((mine()[0].double() - ref).abs().max()) / ref.abs().max())
    print(f"M={M:6d} N={N:4d} K={K:4d}: library fp32+sum {t1:7.1f} us | library 3xbf16+sum {t2:7.1f} us | gemm_tn+reduce {t3:7.1f} us  (rel err {err:.1e})")
