#!/usr/bin/env python3
"""Where a stage of gemm_nt3's 8-wave tile spends its cycles (round 6): the instrumented instance (COMBO_NT3_DBG=128) stamps s_memtime at the
top of every stage, behind the window barrier and behind phase 0's wait, and sums the three segments per wave:
    seg 0 = counted vmcnt wait + s_barrier      (waiting for the ring / for the other waves)
    seg 1 = phase 0: 6 ds_read_b128 issued, 8 MFMAs with the 3 DMA pieces between them, lgkmcnt(0)
    seg 2 = phase 1: 4 ds_read_b128 issued, 4 MFMAs with the A split between them, lgkmcnt(0)
    COMBO_NT3_DBG=128 python tools/prof_nt3_stage.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import combo_avs_amd  # noqa: F401,E402
from combo_avs_amd import _lib  # noqa: E402
from combo_avs_amd.ops.linear import gemm_nt_x3, presplit  # noqa: E402

assert os.environ.get("COMBO_NT3_DBG") == "128", "run with COMBO_NT3_DBG=128"
lib = _lib.lib()
lib.combo_gemm_nt_x3_tile(1)
torch.manual_seed(0)
for M, K, N in [(32768, 256, 256), (32768, 1024, 256), (8192, 256, 1024), (32768, 2048, 256)]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    img = presplit(w)
    for _ in range(20):
        gemm_nt_x3(a, w, img=img)
    buf = torch.zeros(256 * 8, 4, dtype=torch.int64, device="cuda")
    lib.combo_gemm_nt_x3_prof_buffer(buf.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    gemm_nt_x3(a, w, img=img)
    e.record()
    torch.cuda.synchronize()
    lib.combo_gemm_nt_x3_prof_buffer(None)
    b = buf.cpu().double()
    st = b[:, 3].clamp_min(1)
    seg = b[:, :3] / st[:, None]
    first, second = seg.view(256, 8, 3)[:, :4].mean((0, 1)), seg.view(256, 8, 3)[:, 4:].mean((0, 1))
    tot = seg.sum(1)
    print(f"[{M}x{K}->{N}] {s.elapsed_time(e) * 1e3:.1f} us (instrumented), {int(st.mean())} stages per wave: cycles per stage {tot.mean():.0f} "
          f"(min {tot.min():.0f}, max {tot.max():.0f}) = wait+barrier {seg[:, 0].mean():.0f} + phase 0 {seg[:, 1].mean():.0f} + phase 1 {seg[:, 2].mean():.0f}"
          f" | waves 0-3: {first[0]:.0f} / {first[1]:.0f} / {first[2]:.0f}, waves 4-7: {second[0]:.0f} / {second[1]:.0f} / {second[2]:.0f}", flush=True)
