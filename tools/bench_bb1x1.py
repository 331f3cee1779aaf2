#!/usr/bin/env python3
"""ResNet-50 backbone 1x1 convolutions (bf16, channels_last, B = 40 frames): MIOpen (exhaustive find) vs the same op as a
token-major GEMM through hipBLASLt (x[B*H*W, Cin] @ w[Cout, Cin]^T), forward and backward (dx + dw)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


B = 40
tot = [0.0, 0.0, 0.0, 0.0]
# (H, Cin, Cout, count per backbone)
for H, cin, cout, cnt in [(56, 64, 64, 1), (56, 64, 256, 4), (56, 256, 64, 2), (56, 256, 128, 1), (28, 128, 512, 4), (28, 512, 128, 3),
                          (28, 512, 256, 1), (14, 256, 1024, 6), (14, 1024, 256, 5), (14, 1024, 512, 1), (7, 512, 2048, 3), (7, 2048, 512, 2)]:
    x = torch.randn(B, cin, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device="cuda", dtype=torch.bfloat16) * 0.05).requires_grad_(True)
    y = F.conv2d(x, w)
    g = torch.randn_like(y)
    t_cf = timeit(lambda: F.conv2d(x, w))
    t_cb = timeit(lambda: torch.autograd.grad(y, (x, w), g, retain_graph=True))
    xt = x.detach().permute(0, 2, 3, 1).reshape(-1, cin).requires_grad_(True)
    w2 = w.detach().view(cout, cin).requires_grad_(True)
    y2 = F.linear(xt, w2)
    g2 = g.permute(0, 2, 3, 1).reshape(-1, cout)
    t_gf = timeit(lambda: F.linear(xt, w2))
    t_gb = timeit(lambda: torch.autograd.grad(y2, (xt, w2), g2, retain_graph=True))
    err = float((y.permute(0, 2, 3, 1).reshape(-1, cout).float() - y2.float()).abs().max())
    print(f"{H:3d}x{H:<3d} {cin:4d}->{cout:4d} x{cnt}: MIOpen fwd {t_cf:6.1f} bwd {t_cb:6.1f} | GEMM fwd {t_gf:6.1f} bwd {t_gb:6.1f} us   max diff {err:.3f}", flush=True)
    tot[0] += cnt * t_cf; tot[1] += cnt * t_cb; tot[2] += cnt * t_gf; tot[3] += cnt * t_gb
print(f"per backbone: MIOpen fwd {tot[0]/1e3:.2f} + bwd {tot[1]/1e3:.2f} ms | GEMM fwd {tot[2]/1e3:.2f} + bwd {tot[3]/1e3:.2f} ms")
