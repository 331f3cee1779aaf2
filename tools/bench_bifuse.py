#!/usr/bin/env python3
"""Micro-benchmark of the bilateral-fusion token stage (csrc/bifuse.hip) at the bench's shape (40 frames x 56 x 56 tokens x 256
channels, dropout 0.1): forward and forward + backward per call.  Per-kernel times: run under
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bf -o b -- python3 tools/bench_bifuse.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa: F401
from combo_avs_amd.ops import bifuse

torch.manual_seed(0)
B, N, C, H = 40, 3136, 256, 8
dev = "cuda"
x = torch.randn(B, N, C, device=dev, requires_grad=True)
ln_w, ln_b = (torch.randn(C, device=dev) * 0.1 + 1).requires_grad_(), (torch.randn(C, device=dev) * 0.1).requires_grad_()
pos = torch.randn(1, N, C, device=dev)
u, z = torch.randn(B, H, C, device=dev, requires_grad=True), torch.randn(B, H, C, device=dev, requires_grad=True)
c = torch.randn(B, H, device=dev, requires_grad=True)
b_ov, gam = torch.randn(C, device=dev, requires_grad=True), (torch.randn(C, device=dev) * 0.1).requires_grad_()


def fwd():
    return bifuse.token_op(x, ln_w, ln_b, 1e-5, pos, u, c, z, b_ov, gam, 0.1, seed=7)


def fwd_bwd():
    y, pooled, spa = fwd()
    (y.sum() + pooled.sum() + spa.sum()).backward()


def t(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


with torch.no_grad():
    f = min(t(fwd) for _ in range(3))
fb = min(t(fwd_bwd) for _ in range(3))
print(f"bifuse token stage B={B} N={N}: forward {f:.1f} us, forward + backward {fb:.1f} us (incl. the autograd glue)")
