import torch
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def rel(a, b): return ((a.double() - b).norm() / b.norm()).item()
M, K, N = 41160, 256, 1024
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; dy = torch.randn(M, N, device="cuda")
ref_f = x.double() @ w.double().t(); ref_dx = dy.double() @ w.double(); ref_dw = dy.double().t() @ x.double()
for tf in (False, True):
    torch.backends.cuda.matmul.allow_tf32 = tf
    print("allow_tf32", tf, "fwd %.0f us err %.2e | dX %.0f us err %.2e | dW %.0f us err %.2e" % (
        t(lambda: x @ w.t()), rel(x @ w.t(), ref_f), t(lambda: dy @ w), rel(dy @ w, ref_dx), t(lambda: dy.t() @ x), rel(dy.t() @ x, ref_dw)))
import os
print({k: v for k, v in os.environ.items() if "TF32" in k or "BLAS" in k})
