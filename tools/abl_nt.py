#!/usr/bin/env python3
"""Ablation of csrc/gemm_nt.hip (env COMBO_NT_DBG selects the build): launches through the C ABI into a preallocated output
so that the host costs ~5 us per launch and the loop is GPU-bound.  Usage: COMBO_NT_DBG=<bits> python tools/abl_nt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd import _lib

L = _lib.lib()
st = _lib.current_stream()
dbg = os.environ.get("COMBO_NT_DBG", "0")
res = []
for M, K, N in [(41160, 256, 256), (41160, 256, 1024), (41160, 1024, 256), (125440, 256, 256), (125440, 2304, 256)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    def run():
        L.combo_gemm_nt_x3_f32(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N, M, N, K, 0, st)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 40 * 1e3
    res.append("%dx%dx%d %.1f us (%.0f TF/s bf16)" % (M, K, N, us, 6.0 * M * N * K / us * 1e-6))
print("dbg=%s: " % dbg + " | ".join(res))
