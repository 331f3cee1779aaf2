#!/usr/bin/env python3
"""csrc/gemm_nt.hip (v1; env COMBO_NT_DBG selects an ablation build) and csrc/gemm_nt2.hip (v2) through the C ABI into a
preallocated output, so that the host costs ~5 us per launch and the loop is GPU-bound.
Usage: [COMBO_NT_DBG=<bits>] [COMBO_NT2_STAGGER=0|1] python tools/abl_nt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd import _lib
from combo_avs_amd.ops.linear import presplit

L = _lib.lib()
st = _lib.current_stream()
dbg = os.environ.get("COMBO_NT_DBG", "0")


def timeit(run, n=40):
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


res = []
import torch.nn.functional as F
for M, K, N in [(41160, 256, 256), (41160, 256, 1024), (125440, 2304, 256), (4000, 256, 256), (4000, 256, 768), (4000, 256, 2048), (4000, 2048, 256), (4000, 256, 512), (1960, 256, 256), (7840, 256, 256), (7840, 256, 512)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda"); out2 = torch.empty(M, N, device="cuda")
    img = presplit(w)
    t1 = timeit(lambda: L.combo_gemm_nt_x3_f32(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N, M, N, K, 0, st))
    t2 = timeit(lambda: L.combo_gemm_nt_x3_pre_f32(a.data_ptr(), K, img.data_ptr(), b.data_ptr(), out2.data_ptr(), N, M, N, K, 0, st))
    tp = timeit(lambda: presplit(w))
    torch.backends.cuda.matmul.allow_tf32 = True
    tl = timeit(lambda: F.linear(a, w, b))
    torch.backends.cuda.matmul.allow_tf32 = False
    err = float((out - out2).abs().max())
    res.append("%dx%dx%d v1 %.1f us | v2 %.1f us (%.0f TF/s bf16) presplit %.1f us, hipBLASLt-3x (torch) %.1f us, max|v1-v2| %.1e" % (M, K, N, t1, t2, 6.0 * M * N * K / t2 * 1e-6, tp, tl, err))
print("dbg=%s stagger=%s:\n  " % (dbg, os.environ.get("COMBO_NT2_STAGGER", "1")) + "\n  ".join(res))
