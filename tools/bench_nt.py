#!/usr/bin/env python3
"""gemm_nt_x3 vs hipBLASLt 3xbf16 on the head's forward/dX shapes (to place the dispatch threshold in ops/linear.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd.ops.linear import gemm_nt_x3


def timeit(fn, iters=30):
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


for M, K, N in [(41160, 256, 256), (31360, 256, 256), (7840, 256, 256), (4000, 256, 256), (4000, 256, 2048), (4000, 2048, 256),
                (4000, 256, 768), (31360, 256, 512), (7840, 256, 512), (41160, 256, 288), (41160, 288, 256), (125440, 256, 256)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")

    def lib3():
        torch.backends.cuda.matmul.allow_tf32 = True
        y = torch.nn.functional.linear(a, w, b)
        torch.backends.cuda.matmul.allow_tf32 = False
        return y
    tiles = -(-M // 256) * -(-N // 128)
    print(f"M={M:6d} K={K:4d} N={N:4d} tiles={tiles:4d}: hipBLASLt-3x {timeit(lib3):6.1f} us | gemm_nt_x3 {timeit(lambda: gemm_nt_x3(a, w, b)):6.1f} us")
