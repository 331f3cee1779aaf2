#!/usr/bin/env python3
"""How long do the weight-gradient (wrw) passes of the ResNet-50 1x1 / 3x3 convolutions take in MIOpen (bf16, NHWC,
B = 40 frames)?  Sum per class, with and without MIOpen's exhaustive find."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd.backbone import ResNet, ConvBN

torch.backends.cudnn.benchmark = os.environ.get("FIND", "1") == "1"
m = ResNet(50).cuda()
shapes = []
hooks = []
for name, mod in m.named_modules():
    if isinstance(mod, ConvBN):
        hooks.append(mod.register_forward_hook(lambda mod, inp, out, name=name: shapes.append((name, mod, tuple(inp[0].shape), tuple(out.shape)))))
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    m(torch.randn(40, 3, 224, 224, device="cuda"))
for h in hooks:
    h.remove()


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot = {}
for name, mod, ishape, oshape in shapes:
    x = torch.randn(ishape, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(oshape, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = mod.weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    k, st = mod.kernel_size[0], mod.stride[0]
    f = lambda mask: torch.ops.aten.convolution_backward(gy, x, w, [mod.out_channels], mod.stride, mod.padding, (1, 1), False, (0, 0), 1, mask)
    t_w = timeit(lambda: f([False, True, False]))
    t_x = timeit(lambda: f([True, False, False]))
    t_f = timeit(lambda: torch.nn.functional.conv2d(x, w, None, mod.stride, mod.padding))
    key = f"{k}x{k} s{st}"
    a = tot.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += t_f; a[2] += t_x; a[3] += t_w
    if os.environ.get("VERBOSE"):
        print(f"{name:22s} {key} in {ishape} out {oshape}: fwd {t_f:6.1f} bwd-data {t_x:6.1f} wrw {t_w:6.1f} us")
for key, (n, tf, tx, tw) in sorted(tot.items()):
    print(f"{key}: {n:2d} convs  fwd {tf / 1e3:6.2f} ms  bwd-data {tx / 1e3:6.2f} ms  wrw {tw / 1e3:6.2f} ms")
