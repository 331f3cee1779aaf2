#!/usr/bin/env python3
"""What clock / power does the chip hold under (a) hipBLASLt bf16 GEMMs, (b) gemm_nt v1/v2, (c) v1 without MFMA?  Samples
rocm-smi from a side thread while one kernel is launched back-to-back for ~1.5 s."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd import _lib
from combo_avs_amd.ops.linear import presplit

L = _lib.lib(); st = _lib.current_stream()


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l.strip() for l in r.splitlines() if ("sclk" in l or "Power" in l or "fclk" in l or "mclk" in l)]
            out.append(" ; ".join(x.split(":", 1)[-1].strip()[-40:] for x in s))
        except Exception as e:
            out.append(repr(e))
        time.sleep(0.25)


def run(name, fn, flops, secs=1.5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0; t0 = time.time(); s.record()
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        n += 20
        torch.cuda.synchronize()
    e.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = s.elapsed_time(e) / n * 1e3
    print("%-34s %8.1f us  %7.0f TF/s   smi: %s" % (name, us, flops / us * 1e-6, out[len(out) // 2] if out else "-"), flush=True)


a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
run("hipBLASLt bf16 8192^3", lambda: torch.matmul(a, b), 2 * 8192 ** 3)
M, K, N = 125440, 2304, 256
a2 = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b2 = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
run("hipBLASLt bf16 125440x2304x256 NT", lambda: torch.matmul(a2, b2.t()), 2.0 * M * N * K)
M2, K2, N2 = 41160, 256, 1024
a3 = torch.randn(M2, K2, device="cuda", dtype=torch.bfloat16); b3 = torch.randn(N2, K2, device="cuda", dtype=torch.bfloat16)
run("hipBLASLt bf16 41160x256x1024 NT", lambda: torch.matmul(a3, b3.t()), 2.0 * M2 * N2 * K2)
af = torch.randn(M, K, device="cuda"); wf = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda"); img = presplit(wf)
run("gemm_nt v2 125440x2304x256 (x3)", lambda: L.combo_gemm_nt_x3_pre_f32(af.data_ptr(), K, img.data_ptr(), None, out.data_ptr(), N, M, N, K, 0, st), 6.0 * M * N * K)
x = torch.randn(1 << 28, device="cuda"); y = torch.empty_like(x)
run("copy 1 GiB (GB/s in TF col /1e3)", lambda: y.copy_(x), 2.0 * x.numel() * 4)
