#!/usr/bin/env python3
"""Routing sweep for ops/linear.py::_nt_ok: gemm_nt2 (wide / skinny tiles, incl. its weight pre-split launch) against
hipBLASLt's 3xbf16 mode on the GEMM shapes of the head, every candidate timed INSIDE a captured hipGraph (20 launches per
replay) so that host overhead does not hide the difference.  COMBO_NT2_SKINNY=0|2 forces the tile configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import combo_avs_amd
from combo_avs_amd import _lib
from combo_avs_amd.ops.linear import presplit

L = _lib.lib()


def graph_time(fn, reps=20, replays=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * replays) * 1e3


for M, K, N in [(31360, 256, 256), (7840, 256, 256), (1960, 256, 256), (4000, 256, 256), (4000, 256, 768), (4000, 256, 2048),
                (4000, 2048, 256), (41160, 256, 96), (41160, 256, 288), (31360, 256, 512), (7840, 256, 512), (16384, 2048, 256)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    img = presplit(w)

    def nt2():
        st = _lib.current_stream()
        L.combo_presplit_bf16x2_f32(w.data_ptr(), K, 1, N, K, img.data_ptr(), st)
        L.combo_gemm_nt_x3_pre_f32(a.data_ptr(), K, img.data_ptr(), b.data_ptr(), out.data_ptr(), N, M, N, K, 0, st)

    def lib3():
        return F.linear(a, w, b)
    torch.backends.cuda.matmul.allow_tf32 = True
    t_lib = graph_time(lib3)
    torch.backends.cuda.matmul.allow_tf32 = False
    t_nt = graph_time(nt2)
    tiles = -(-M // 256) * -(-N // 128)
    print(f"M={M:6d} K={K:4d} N={N:4d} wide tiles={tiles:4d}: hipBLASLt-3x {t_lib:6.1f} us | gemm_nt2 + presplit {t_nt:6.1f} us", flush=True)
