#!/usr/bin/env python3
"""Depth-wise 3x3 weight gradient (csrc/dwconv.hip): the strip form (round 5) against the per-token form (round 3) on the PVTv2-B5
MLP shapes of a 40-frame 224^2 batch and an 80-frame 512^2 batch; same partial layout, same finish kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa: F401
from combo_avs_amd import _lib
lib = _lib.lib()


def run(x, dy, B, H, W, C):
    slices = lib.combo_dwconv3x3_wgrad_slices(B, H, W, C)
    part = torch.empty(slices, 10, C, device="cuda", dtype=torch.float32)
    dw = torch.empty(C, 9, device="cuda"); db = torch.empty(C, device="cuda")
    st = _lib.current_stream()
    def f():
        _lib.check(lib.combo_dwconv3x3_wgrad_bf16(x.data_ptr(), dy.data_ptr(), B, H, W, C, slices, part.data_ptr(), st), "wgrad")
        _lib.check(lib.combo_dwconv3x3_wgrad_finish_f32(part.data_ptr(), slices, C, dw.data_ptr(), db.data_ptr(), st), "finish")
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 50 * 1e3, dw.clone(), db.clone(), slices


for (B, H, W, C) in [(40, 56, 56, 256), (40, 28, 28, 512), (40, 14, 14, 1280), (40, 7, 7, 2048), (80, 128, 128, 256), (80, 64, 64, 512),
                     (80, 32, 32, 1280), (80, 16, 16, 2048)]:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    lib.combo_dwconv3x3_wgrad_strips(0)
    t2, dw2, db2, s2 = run(x, dy, B, H, W, C)
    lib.combo_dwconv3x3_wgrad_strips(1)
    t3, dw3, db3, s3 = run(x, dy, B, H, W, C)
    mb = 2 * x.numel() * 2 / 1e6
    print(f"[{B}x{H}x{W}x{C}] per-token {t2:7.1f} us ({s2} slices)  strips {t3:7.1f} us ({s3} slices)  {mb:.0f} MB read once = {mb / t3 / 1e3 * 1e3:.0f} GB/s"
          f"  max |dw diff| {float((dw2 - dw3).abs().max()):.2e} of {float(dw2.abs().max()):.2e}, |db diff| {float((db2 - db3).abs().max()):.2e}", flush=True)
