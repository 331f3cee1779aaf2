#!/usr/bin/env python3
"""Micro-benchmark of the depth-wise 3x3 kernels (csrc/dwconv.hip) at PVTv2-B5's MLP shapes: forward per launch against the
bytes of one read + one write of the activation.  usage: python tools/bench_dwconv.py [frames] [image side] (default 80, 512)
HBM traffic: rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --kernel-trace --output-format csv -d DIR -o p -- python3 tools/bench_dwconv.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa: F401
from combo_avs_amd.ops.dwconv import dwconv3x3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 80
side = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = int(os.environ.get("ITERS", "10"))
for stride, C in ((4, 256), (8, 512), (16, 1280), (32, 2048)):
    H = W = side // stride
    x = torch.randn(B, H, W, C, device="cuda").bfloat16()
    w = torch.randn(C, 1, 3, 3, device="cuda") * 0.3
    b = torch.randn(C, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            dwconv3x3(x, w, b)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            dwconv3x3(x, w, b)
        e.record()
        torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    gb = x.numel() * 4 / 1e9
    print(f"dwconv3x3 [{B},{H},{W},{C}] bf16: {us:.1f} us, {gb / us * 1e3:.2f} TB/s of one read + one write ({gb * 1e3:.0f} MB)", flush=True)
