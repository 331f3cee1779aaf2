#!/usr/bin/env python3
"""What clock / power does the chip hold under (a) gemm_nt3's wide tile (3 bf16 products), (b) the same on fp16 pieces, (c) the exact fp32
MFMA kernel, (d) an HBM copy?  Samples rocm-smi from a side thread while one kernel is launched back to back for ~1.5 s (round 6)."""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import combo_avs_amd  # noqa: F401,E402
from combo_avs_amd.ops import linear as L  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [ln.strip() for ln in r.splitlines() if ("sclk" in ln or "Power" in ln)]
            out.append(" ; ".join(x.split(":", 1)[-1].strip()[-44:] for x in s))
        except Exception as e:  # noqa: BLE001
            out.append(repr(e))
        time.sleep(0.2)


def run(name, fn, flops, secs=1.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    s.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = s.elapsed_time(e) / n * 1e3
    print(f"[{name}] {us:.1f} us per launch = {flops / us / 1e6:.1f} TFLOP/s useful; rocm-smi samples (last 4): {out[-4:]}", flush=True)


M, K, N = 32768, 1024, 256
a = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.05
img = L.presplit(w)
run("gemm_nt3 wide, 3 bf16 products", lambda: L.gemm_nt_x3(a, w, img=img), 2.0 * M * N * K)
L.set_forward_precision("f16x3")
img16 = L.presplit(w, True)
run("gemm_nt3 wide, 3 fp16 products", lambda: L.gemm_nt_bf16(a, w, img=img16), 2.0 * M * N * K)
L.set_forward_precision("fp32")
run("gemm_nt_f32 (exact fp32 MFMA)", lambda: L.gemm_nt_f32(a, w), 2.0 * M * N * K)
x = torch.randn(64 << 20, device="cuda")
y = torch.empty_like(x)
run("HBM copy 256 MB", lambda: y.copy_(x), 0.0)
