#!/usr/bin/env python3
"""Clock / package power the chip holds under the exact-fp32 MFMA GEMM (csrc/gemm_f32.hip), its pure-MFMA ablation
(COMBO_F32_DBG=15 in the environment) and the library's fp32 GEMM: rocm-smi sampled from a side thread while one kernel is launched
back-to-back for ~2 s.  Run once per COMBO_F32_DBG value (the ablation bits are read once per process)."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa: F401
from combo_avs_amd.ops.linear import gemm_nt_f32


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [ln.strip() for ln in r.splitlines() if ("sclk" in ln or "Power" in ln)]
            out.append(" ; ".join(x.split(":", 1)[-1].strip()[-34:] for x in s))
        except Exception as e:  # noqa: BLE001
            out.append(repr(e))
        time.sleep(0.2)


def run(name, fn, flops, secs=2.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    s.record()
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = s.elapsed_time(e) / n * 1e3
    print("%-44s %8.1f us  %6.1f TF/s   smi: %s" % (name, us, flops / us * 1e-6, " | ".join(out[1:4]) if out else "-"), flush=True)


tag = "DBG=" + os.environ.get("COMBO_F32_DBG", "0") + " TILE=" + os.environ.get("COMBO_F32_TILE", "auto")
for M, K, N in ((41160, 256, 1024), (125440, 256, 256), (8192, 8192, 8192)):
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    out = torch.empty(M, N, device="cuda")
    run(f"gemm_nt_f32 {M}x{K}->{N} [{tag}]", lambda: gemm_nt_f32(a, w, None, False, out=out), 2.0 * M * N * K)
    if os.environ.get("COMBO_F32_DBG", "0") == "0":
        run(f"library fp32 {M}x{K}->{N}", lambda: torch.nn.functional.linear(a, w), 2.0 * M * N * K)
