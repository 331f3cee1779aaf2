import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.backends.cudnn.benchmark = True
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for fmt in (torch.contiguous_format, torch.channels_last):
    x = torch.randn(40, 256, 56, 56, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
    w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, w, None, 1, 1)
    g = torch.randn_like(y)
    print(fmt, "out channels_last:", y.is_contiguous(memory_format=torch.channels_last),
          f"fwd {timeit(lambda: torch.nn.functional.conv2d(x, w, None, 1, 1)):.0f} us",
          f"bwd(dx+dw) {timeit(lambda: torch.autograd.grad(y, (x, w), g, retain_graph=True)):.0f} us")
    gn = torch.nn.GroupNorm(32, 256).cuda()
    yy = y.detach().requires_grad_(True)
    z = gn(yy)
    print("   GroupNorm out channels_last:", z.is_contiguous(memory_format=torch.channels_last), f"fwd {timeit(lambda: gn(yy)):.0f} us",
          f"bwd {timeit(lambda: torch.autograd.grad(z, (yy, gn.weight, gn.bias), g, retain_graph=True)):.0f} us")

# the implicit-GEMM path (ops/conv3x3.py: csrc/gemm_nt.hip / gemm_tn.hip with CONV = true)
import combo_avs_amd  # noqa: E402
from combo_avs_amd.ops import conv3x3 as C  # noqa: E402
x = torch.randn(40, 256, 56, 56, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).requires_grad_(True)
y = C.conv3x3(x, w)
g = torch.randn(40, 256, 56, 56, device="cuda").contiguous(memory_format=torch.channels_last)
ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
print("conv3x3 HIP: max |y - miopen| =", float((y - ref).abs().max()), " fwd %.0f us" % timeit(lambda: C.conv3x3(x, w)),
      " dx %.0f us" % timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True)),
      " dw %.0f us" % timeit(lambda: torch.autograd.grad(y, w, g, retain_graph=True)))
