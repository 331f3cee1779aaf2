#!/usr/bin/env python3
"""30 forward + backward passes of the decoder attention at one key length (for `rocprofv3 --kernel-trace --stats`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa: F401
from combo_avs_amd.ops.attention import attention
B, H, E, Lq = 40, 8, 256, 100
Lk = int(sys.argv[1]) if len(sys.argv) > 1 else 784
torch.manual_seed(0)
q = torch.randn(B * Lq, E, device="cuda", requires_grad=True)
k = torch.randn(B * Lk, E, device="cuda", requires_grad=True)
v = torch.randn(B * Lk, E, device="cuda", requires_grad=True)
pitch = (Lk + 3) // 4 * 4
blocked = (torch.rand(B, Lq, pitch, device="cuda") < 0.5).to(torch.uint8)
blocked[:, :, 0] = 0
g = torch.randn(B * Lq, E, device="cuda")
for _ in range(30):
    torch.autograd.grad(attention(q, k, v, blocked, B, H), (q, k, v), g)
torch.cuda.synchronize()
print("done", flush=True)
