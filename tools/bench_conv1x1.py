import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd.ops.linear import linear
torch.backends.cudnn.benchmark = True
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, Cin, Cout, H) in [(40, 256, 256, 56), (40, 512, 256, 28), (40, 1024, 256, 14), (40, 2048, 256, 7)]:
    x = torch.randn(B, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(Cout, device="cuda").requires_grad_(True)
    y = torch.nn.functional.conv2d(x, w, b)
    g = torch.randn_like(y)
    t_f = timeit(lambda: torch.nn.functional.conv2d(x, w, b))
    t_b = timeit(lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True))
    xt = x.detach().permute(0, 2, 3, 1).reshape(-1, Cin).requires_grad_(True)
    w2 = w.detach().view(Cout, Cin).clone().requires_grad_(True)
    y2 = linear(xt, w2, b)
    g2 = torch.randn_like(y2)
    t_f2 = timeit(lambda: linear(xt, w2, b))
    t_b2 = timeit(lambda: torch.autograd.grad(y2, (xt, w2, b), g2, retain_graph=True))
    print(f"1x1 conv {Cin}->{Cout} @ {H}x{H} x{B}: MIOpen fp32 fwd {t_f:.0f} us bwd {t_b:.0f} us | linear (gemm_nt/tn) fwd {t_f2:.0f} us bwd {t_b2:.0f} us")
