#!/usr/bin/env python3
"""The stride-1 convolutions of the fp32 ResNet-50 backbone at BT = 40 on the head's 3-product kernel (csrc/gemm_nt3.hip: 1x1 as
a token GEMM, 3x3 as the implicit GEMM) with the bias + ReLU epilogue, next to the library's convolution followed by the
separate bias/ReLU pass the step runs today (ops/biasact.py).  Answers "what would the own forward path buy"."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import combo_avs_amd  # noqa: E402,F401
from combo_avs_amd import _lib  # noqa: E402
from combo_avs_amd.ops import linear as L  # noqa: E402
from combo_avs_amd.ops.biasact import bias_act  # noqa: E402

torch.backends.cudnn.benchmark = True
BT = 40
LAYERS = []
inp, H = 64, 56
for si, (mid, out, nblk, stride) in enumerate([(64, 256, 3, 1), (128, 512, 4, 2), (256, 1024, 6, 2), (512, 2048, 3, 2)]):
    s = f"res{si + 2}"
    Hout = H // stride
    LAYERS += [(f"{s}.0 conv1 1x1", 1, inp, mid, 1, H), (f"{s}.x conv3 1x1 (+res)", nblk, mid, out, 1, Hout),
               (f"{s}.x conv1 1x1", nblk - 1, out, mid, 1, Hout), (f"{s}.x conv2 3x3", nblk - 1 + (stride == 1), mid, mid, 3, Hout)]
    if stride == 1:
        LAYERS.append((f"{s}.0 shortcut 1x1", 1, inp, out, 1, H))
    inp, H = out, Hout


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


lib = _lib.lib()
tot = [0.0, 0.0]
print(f"{'layer':28s} {'x':>2s} {'GF':>6s} | {'library+bias_act us':>20s} | {'own x3 us':>10s} {'TF/s':>6s} {'GB/s':>6s} | max rel err")
for name, cnt, cin, cout, k, Hin in LAYERS:
    x = torch.randn(BT, cin, Hin, Hin, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * (cin * k * k) ** -0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda")
    M = BT * Hin * Hin
    x_tok = x.permute(0, 2, 3, 1).reshape(M, cin)
    wm = w.permute(0, 2, 3, 1).reshape(cout, k * k * cin)
    img = L.presplit(wm)
    y = torch.empty(M, cout, device="cuda")
    st = _lib.current_stream()

    def own():
        if k == 1:
            rc = lib.combo_gemm_nt_x3_pre_f32(x_tok.data_ptr(), cin, img.data_ptr(), b.data_ptr(), y.data_ptr(), cout, M, cout, cin, 1, st)
        else:
            rc = lib.combo_conv3x3_nhwc_x3_pre_f32(x_tok.data_ptr(), cin, img.data_ptr(), b.data_ptr(), y.data_ptr(), cout, BT, Hin, Hin, cin,
                                                   cout, 1, st)
        assert rc == 0

    def library():
        return bias_act(torch.nn.functional.conv2d(x, w, None, 1, k // 2), b)
    ref = library().permute(0, 2, 3, 1).reshape(M, cout).double()
    own()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    t_l, t_o = timeit(library), timeit(own)
    gf = 2.0 * M * cout * cin * k * k / 1e9
    gb = 4.0 * (M * cin + M * cout + cout * cin * k * k) / 1e9
    tot[0] += cnt * t_l
    tot[1] += cnt * t_o
    print(f"{name:28s} {cnt:2d} {gf:6.2f} | {t_l:20.1f} | {t_o:10.1f} {gf / t_o * 1e3:6.1f} {gb / t_o * 1e6:6.0f} | {err:.1e}", flush=True)
print(f"per backbone, stride-1 layers: library + bias/ReLU pass {tot[0] / 1e3:.2f} ms, own 3-product kernels {tot[1] / 1e3:.2f} ms")
