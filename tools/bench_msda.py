#!/usr/bin/env python3
"""Micro-benchmark of the MSDeformAttn core kernels at the BASELINE config-2 shape (BT=40, S=1029, M=8, D=32).
Reports average kernel time (HIP events) and algorithmic GB/s: 3.29 MB per frame-layer fwd."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa
from combo_avs_amd import msda

ap = argparse.ArgumentParser()
ap.add_argument("--bt", type=int, default=40)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--spread", type=float, default=2.5, help="offset std in pixels")
a = ap.parse_args()
torch.manual_seed(0)
dev = "cuda:0"
shapes = [(7, 7), (14, 14), (28, 28)]
B, S, M, D, L, P = a.bt, 1029, 8, 32, 3, 4
value = torch.randn(B, S, M, D, device=dev)
ref = torch.rand(B, S, 1, 1, 1, 2, device=dev)
norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=dev)
loc = (ref + torch.randn(B, S, M, L, P, 2, device=dev) * a.spread / norm[None, None, None, :, None, :]).contiguous()
w = torch.softmax(torch.randn(B, S, M, L * P, device=dev), -1).view(B, S, M, L, P).contiguous()
go = torch.randn(B, S, M * D, device=dev)
sh = torch.as_tensor(shapes, dtype=torch.int64, device=dev)
lsi = torch.tensor([0, 49, 245], device=dev)

fwd_bytes = B * (S * M * D * 4 * 2 + S * M * L * P * 3 * 4)           # value + out + loc + w
bwd_bytes = B * (S * M * D * 4 * 3 + S * M * L * P * 3 * 4 * 2)       # gout, value, gvalue + loc,w + gloc,gw


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


for algo, name in ((1, "generic"), (2, "lds"), (3, "tap")):
    msda.set_algo(algo)
    t = timeit(lambda: msda.ms_deform_attn_forward(value, sh, lsi, loc, w), a.iters)
    print(f"fwd {name:8s}: {t:8.1f} us   {fwd_bytes / t / 1e3:8.1f} GB/s algorithmic ({fwd_bytes/1e6:.1f} MB)")
    t = timeit(lambda: msda.ms_deform_attn_backward(value, sh, lsi, loc, w, go), a.iters)
    print(f"bwd {name:8s}: {t:8.1f} us   {bwd_bytes / t / 1e3:8.1f} GB/s algorithmic ({bwd_bytes/1e6:.1f} MB) (incl. 3 memsets)")

# fused windowed backward (csrc/msda_bwd.hip) against the two-kernel LDS path, same inputs; max deviation between the two
msda.set_algo(0)
for flag, name in ((False, "two-kernel"), (True, "windowed")):
    msda.WINDOWED_BACKWARD = flag
    t = timeit(lambda: msda.ms_deform_attn_backward(value, sh, lsi, loc, w, go), a.iters)
    print(f"bwd {name:10s}: {t:8.1f} us   {bwd_bytes / t / 1e3:8.1f} GB/s algorithmic ({bwd_bytes/1e6:.1f} MB)")
msda.WINDOWED_BACKWARD = False
r0 = msda.ms_deform_attn_backward(value, sh, lsi, loc, w, go)
msda.WINDOWED_BACKWARD = True
r1 = msda.ms_deform_attn_backward(value, sh, lsi, loc, w, go)
r2 = msda.ms_deform_attn_backward(value, sh, lsi, loc, w, go)
for nm, x, y, z in zip(("grad_value", "grad_loc", "grad_w"), r0, r1, r2):
    print(f"{nm}: max |windowed - two-kernel| = {(x - y).abs().max().item():.3e} (max |x| {x.abs().max().item():.3e}); "
          f"windowed run-to-run bitwise equal: {bool(torch.equal(y, z))}")
