import torch
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, K, N) in [(41160, 256, 1024), (41160, 1024, 256), (41160, 256, 256), (4000, 256, 2048), (31360, 256, 256)]:
    x = torch.randn(M, K, device="cuda"); dy = torch.randn(M, N, device="cuda")
    xt = x.t().contiguous(); dyt = dy.t().contiguous()
    res = []
    for tf in (False, True):
        torch.backends.cuda.matmul.allow_tf32 = tf
        res.append((tf,
            t(lambda: dy.t() @ x),                      # TN as autograd does
            t(lambda: (x.t() @ dy).t()),                # other orientation
            t(lambda: dyt @ x),                         # NN with pre-transposed dY
            t(lambda: dy.t().contiguous() @ x),         # incl. the transpose copy
            t(lambda: dyt @ xt.t()),                    # NT with both transposed
            t(lambda: torch.einsum("mn,mk->nk", dy, x)),
        ))
    for r in res:
        print(f"{M}x{K}x{N} tf32={r[0]}: dy.T@x {r[1]:.0f} | (x.T@dy).T {r[2]:.0f} | dyT_c@x {r[3]:.0f} | transpose+mm {r[4]:.0f} | dyT_c@xT_c.T {r[5]:.0f} | einsum {r[6]:.0f} us")
