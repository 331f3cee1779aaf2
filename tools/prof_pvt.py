#!/usr/bin/env python3
"""Kernel inventory of one PVTv2-B5 backbone forward+backward (bf16 autocast, B = 40 frames, 224x224)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd.backbone_pvt import PyramidVisionTransformerV2
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
m = PyramidVisionTransformerV2(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), qkv_bias=True, norm_eps=1e-6,
                               depths=(3, 6, 40, 3), drop_path_rate=0.1).cuda().train()
x = torch.randn(int(os.environ.get("B", "40")), 3, 224, 224, device="cuda")
params = [p for p in m.parameters()]


def run():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=os.environ.get("DTYPE", "bf16") == "bf16"):
        out = m(x)
    loss = sum(v.float().mean() for v in out.values())
    torch.autograd.grad(loss, params)


for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.self_device_time_total > 0]
print(f"launches={sum(e.count for e in ev)} gpu_ms={sum(e.self_device_time_total for e in ev) / 1e3:.2f}")
for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:22]:
    print(f"  {e.self_device_time_total / 1e3:8.3f} ms {e.count:5d}x  {e.key[:120]}")
