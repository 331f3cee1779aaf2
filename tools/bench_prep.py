import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd.ops.linear import linear_cat, linear
from combo_avs_amd.ops.msdaprep import msda_prep
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, Lq, C, M, L, P = 40, 1029, 256, 8, 3, 4
q = torch.randn(B, Lq, C, device="cuda", requires_grad=True)
w1 = (torch.randn(192, C, device="cuda") * 0.05).requires_grad_(True); b1 = torch.randn(192, device="cuda").requires_grad_(True)
w2 = (torch.randn(96, C, device="cuda") * 0.05).requires_grad_(True); b2 = torch.randn(96, device="cuda").requires_grad_(True)
ref = torch.rand(1, Lq, L, 2, device="cuda").expand(B, -1, -1, -1)
norm = torch.tensor([[7.0, 7.0], [14.0, 14.0], [28.0, 28.0]], device="cuda")
proj = torch.randn(B, Lq, 288, device="cuda", requires_grad=True)
g1 = torch.randn(B, Lq, M, L, P, 2, device="cuda"); g2 = torch.randn(B, Lq, M, L, P, device="cuda")
def prep_fwd(): return msda_prep(proj, ref, norm, M, L, P)
loc, attn = prep_fwd()
def prep_bwd(): return torch.autograd.grad([loc, attn], [proj], [g1, g2], retain_graph=True)
def torch_fwd():
    off = proj[..., :192].reshape(B, Lq, M, L, P, 2); lg = proj[..., 192:].reshape(B, Lq, M, L * P)
    return ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :], lg.softmax(-1)
print(f"prep fwd {timeit(prep_fwd):.1f} us, prep bwd {timeit(prep_bwd):.1f} us, torch fwd (incl. slicing copies) {timeit(torch_fwd):.1f} us")
def fused(): return linear_cat(q, w1, b1, w2, b2)
def sep(): return linear(q, w1, b1), linear(q, w2, b2)
print(f"linear_cat fwd {timeit(fused):.1f} us vs two linears {timeit(sep):.1f} us")
y = fused(); gy = torch.randn_like(y)
ya, yb = sep(); ga, gb = torch.randn_like(ya), torch.randn_like(yb)
print(f"linear_cat bwd {timeit(lambda: torch.autograd.grad(y, [q, w1, b1, w2, b2], gy, retain_graph=True)):.1f} us vs two linears bwd "
      f"{timeit(lambda: torch.autograd.grad([ya, yb], [q, w1, b1, w2, b2], [ga, gb], retain_graph=True)):.1f} us")
