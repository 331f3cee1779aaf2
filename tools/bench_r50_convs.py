#!/usr/bin/env python3
"""What the library (MIOpen / CK through PyTorch) spends on every distinct convolution of the fp32 ResNet-50 backbone at BT = 40
(224 x 224, channels_last): forward, input gradient, weight gradient - timed apart, with the layer's multiplicity per backbone.
Answers "which library convolutions are worth an own kernel" (the step runs two backbones)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.backends.cudnn.benchmark = True
BT = 40
# (name, count per backbone, Cin, Cout, k, stride, H_in)
LAYERS = [("stem 7x7/2", 1, 3, 64, 7, 2, 224)]
inp, H = 64, 56
for si, (mid, out, nblk, stride) in enumerate([(64, 256, 3, 1), (128, 512, 4, 2), (256, 1024, 6, 2), (512, 2048, 3, 2)]):
    s = f"res{si + 2}"
    Hout = H // stride
    LAYERS += [(f"{s}.0 shortcut 1x1/{stride}", 1, inp, out, 1, stride, H), (f"{s}.0 conv1 1x1", 1, inp, mid, 1, 1, H),
               (f"{s}.0 conv2 3x3/{stride}", 1, mid, mid, 3, stride, H), (f"{s}.x conv3 1x1", nblk, mid, out, 1, 1, Hout),
               (f"{s}.x conv1 1x1", nblk - 1, out, mid, 1, 1, Hout), (f"{s}.x conv2 3x3", nblk - 1, mid, mid, 3, 1, Hout)]
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


tot = [0.0, 0.0, 0.0]
print(f"{'layer':28s} {'x':>2s} {'GF':>6s} | {'fwd us':>8s} {'TF/s':>6s} | {'dX us':>8s} {'TF/s':>6s} | {'dW us':>8s} {'TF/s':>6s}")
for name, cnt, cin, cout, k, stride, Hin in LAYERS:
    x = torch.randn(BT, cin, Hin, Hin, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    y = torch.nn.functional.conv2d(x, w, None, stride, pad)
    dy = torch.randn_like(y)
    gf = 2.0 * y.numel() * cin * k * k / 1e9
    t_f = timeit(lambda: torch.nn.functional.conv2d(x, w, None, stride, pad))
    t_x = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (stride, stride), (pad, pad), (1, 1), False, (0, 0), 1, (True, False, False)))
    t_w = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (stride, stride), (pad, pad), (1, 1), False, (0, 0), 1, (False, True, False)))
    for i, t in enumerate((t_f, t_x, t_w)):
        tot[i] += cnt * t
    print(f"{name:28s} {cnt:2d} {gf:6.2f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} | {t_x:8.1f} {gf / t_x * 1e3:6.1f} | {t_w:8.1f} {gf / t_w * 1e3:6.1f}", flush=True)
print(f"per backbone: forward {tot[0] / 1e3:.2f} ms, input gradients {tot[1] / 1e3:.2f} ms, weight gradients {tot[2] / 1e3:.2f} ms (x 2 backbones per step)")
