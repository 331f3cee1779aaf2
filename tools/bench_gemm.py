#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, combo_avs_amd
from combo_avs_amd.ops.linear import gemm_x3
shapes = [(41160, 256, 1024), (41160, 1024, 256), (41160, 256, 256), (31360, 256, 256), (4000, 256, 2048)]
for M, K, N in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); dy = torch.randn(M, N, device="cuda")
    def t(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    print(f"{M}x{K}x{N}: fwd lib {t(lambda: torch.nn.functional.linear(x, w)):6.0f} x3 {t(lambda: gemm_x3(x, False, w, False, M, N, K)):6.0f} | "
          f"dX lib {t(lambda: dy @ w):6.0f} x3 {t(lambda: gemm_x3(dy, False, w, True, M, K, N)):6.0f} | "
          f"dW lib {t(lambda: dy.t() @ x):6.0f} x3 {t(lambda: gemm_x3(dy, True, x, True, N, K, M, splits=max(1, min(64, M // 2048)))):6.0f} us")
