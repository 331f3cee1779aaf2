#!/usr/bin/env python3
"""Round 6: the implicit-GEMM weight gradient of the ResNet-50 3x3 / strided layers (csrc/gemm_tn.hip conv3x3_wgrad_kernel +
splitk_reduce_nchw) as a function of the token split count, at B = 40 frames.  The planner (combo_gemm_tn_splits) aims at two workgroups
per CU; the res5 layers (1 960 tokens, 72 output tiles) get 8 splits of 245 tokens - 147 us for 9.25 GFLOP inside the step.
    python tools/bench_conv_wgrad_splits.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa: F401
from combo_avs_amd import _lib

lib = _lib.lib()
B = 40
LAYERS = [("res2 3x3", 56, 64, 64, 3, 1), ("res3 3x3", 28, 128, 128, 3, 1), ("res4 3x3", 14, 256, 256, 3, 1), ("res5 3x3", 7, 512, 512, 3, 1),
          ("res3.0 3x3 s2", 56, 128, 128, 3, 2), ("res4.0 3x3 s2", 28, 256, 256, 3, 2), ("res5.0 3x3 s2", 14, 512, 512, 3, 2),
          ("res5.0 shortcut s2", 14, 1024, 2048, 1, 2)]


def run(x, dy, H, cin, cout, ks, stride, splits, iters):
    Ho = -(-H // stride)
    M, K = B * Ho * Ho, ks * ks * cin
    mchunk = (-(-M // splits) + 15) // 16 * 16
    splits = -(-M // mchunk)
    part = torch.empty(splits, cout, K, device="cuda")
    dw = torch.empty(cout, cin, ks, ks, device="cuda")
    st = _lib.current_stream()

    def once():
        _lib.check(lib.combo_conv_wgrad_x3_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), part.data_ptr(), B, H, H, cin, cout, ks,
                                               stride, splits, st), "wgrad")
        _lib.check(lib.combo_splitk_reduce_nchw_f32(part.data_ptr(), splits, cout, ks * ks, cin, dw.data_ptr(), st), "reduce")
    for _ in range(20):
        once()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        once()
    e.record()
    torch.cuda.synchronize()
    return splits, s.elapsed_time(e) / iters * 1e3, dw


def main():
    torch.manual_seed(0)
    # warm the clocks
    a = torch.randn(8192, 8192, device="cuda")
    for _ in range(20):
        a @ a
    for name, H, cin, cout, ks, stride in LAYERS:
        Ho = -(-H // stride)
        x = torch.randn(B * H * H, cin, device="cuda")
        dy = torch.randn(B * Ho * Ho, cout, device="cuda")
        M, K = B * Ho * Ho, ks * ks * cin
        planned = lib.combo_gemm_tn_splits(M, cout, K)
        res, ref = [], None
        for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            if s > max(1, M // 128):
                continue
            sp, us, dw = run(x, dy, H, cin, cout, ks, stride, s, 40)
            if ref is None:
                ref = dw.clone()
            err = float((dw - ref).abs().max() / ref.abs().max())
            res.append((sp, us, err))
        best = min(res, key=lambda r: r[1])
        gf = 2.0 * M * cout * K / 1e9
        print(f"{name:20s} M {M:6d} [{cout} x {K}] {gf:6.2f} GF  planner {planned:2d}  |  " + "  ".join(f"{sp}:{us:.0f}" for sp, us, _ in res)
              + f"  | best {best[0]} = {best[1]:.0f} us ({gf / best[1] * 1e3:.0f} TF/s), max dev between split counts {max(r[2] for r in res):.1e}")


if __name__ == "__main__":
    main()
