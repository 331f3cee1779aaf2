#!/usr/bin/env python3
"""Error of the forward GEMM modes against an fp64 product on the head's operand statistics (round 6):
exact fp32 (csrc/gemm_f32.hip), "x3" (3 products on bf16 pieces), "f16x3" (3 products on fp16 pieces), both on csrc/gemm_nt3.hip.
    python tools/probe_f16x3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import combo_avs_amd  # noqa: F401,E402
from combo_avs_amd.ops import linear as L  # noqa: E402

torch.manual_seed(0)
for (M, K, N, sa, sw, what) in [(41160, 256, 1024, 1.0, 0.05, "LN output x weight"), (41160, 1024, 256, 3.0, 0.03, "FFN hidden (ReLU) x weight"),
                                (4000, 256, 256, 1.0, 0.06, "decoder projection"), (31360, 512, 256, 30.0, 0.02, "backbone feature x input projection"),
                                (8192, 256, 256, 1e-3, 1e-3, "small operands (fp16 subnormal pieces)"), (8192, 256, 256, 2000.0, 1.0, "large operands")]:
    a = torch.randn(M, K, device="cuda") * sa
    if "ReLU" in what:
        a = a.relu()
    w = torch.randn(N, K, device="cuda") * sw
    b = torch.randn(N, device="cuda") * 0.1
    ref = (a.double() @ w.double().t() + b.double())
    row = []
    for mode in ("fp32", "x3", "f16x3"):
        L.set_forward_precision(mode)
        try:
            y = L.forward_gemm(a, w, b, False)
        finally:
            L.set_forward_precision(L.DEFAULT_FORWARD_PRECISION)
        err = (y.double() - ref)
        row.append(f"{mode}: rel L2 {float(err.norm() / ref.norm()):.3e}, max / RMS {float(err.abs().max() / ref.pow(2).mean().sqrt()):.3e}, mean {float(err.mean() / ref.pow(2).mean().sqrt()):+.1e}")
    print(f"[{M}x{K}->{N}, |a| ~ {sa}, |w| ~ {sw}: {what}]\n    " + "\n    ".join(row), flush=True)
