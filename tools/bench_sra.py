#!/usr/bin/env python3
"""Micro-benchmark of the spatial-reduction attention kernels (csrc/sra_attention.hip) against the library attention they replace,
at the shapes of PVTv2-B5's four stages: BASELINE configs[4] (4 clips x 10 frames, 224 x 224: 49 keys) and configs[3] (8 x 10, 512 x 512:
256 keys).  HIP events around the forward and around forward + backward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa
from combo_avs_amd.ops import sra

SHAPES = [("224 s1", 40, 3136, 1, 49), ("224 s2", 40, 784, 2, 49), ("224 s3", 40, 196, 5, 49), ("224 s4", 40, 49, 8, 49),
          ("512 s1", 80, 16384, 1, 256), ("512 s2", 80, 4096, 2, 256), ("512 s3", 80, 1024, 5, 256), ("512 s4", 80, 256, 8, 256)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, B, N, h, Nk in SHAPES:
    C = 64 * h
    q = torch.randn(B, N, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    kv = torch.randn(B, Nk, 2 * C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    dout = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)

    def own_f():
        return sra.sra_attention(q, kv, h, 0.125)

    def lib_f():
        k, v = kv.view(B, -1, 2, h, 64).unbind(2)
        o = torch.nn.functional.scaled_dot_product_attention(q.view(B, N, h, 64).transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=0.125)
        return o.transpose(1, 2).reshape(B, N, C)

    def fb(f):
        def run():
            torch.autograd.grad(f(), (q, kv), dout)
        return run
    with torch.no_grad():
        of, lf = timeit(own_f), timeit(lib_f)
    ob, lb = timeit(fb(own_f)), timeit(fb(lib_f))
    io = 2.0 * (2 * B * N * C + 2 * B * Nk * C)  # forward bytes: q + out + kv
    fl = 4.0 * B * h * N * Nk * 64
    print(f"[{name}: B={B} N={N} h={h} Nk={Nk}] forward own {of:7.1f} us ({io / of / 1e6:6.2f} TB/s, {fl / of / 1e6:6.1f} TF/s)  library {lf:7.1f} us | "
          f"fwd+bwd own {ob:7.1f} us  library {lb:7.1f} us")
