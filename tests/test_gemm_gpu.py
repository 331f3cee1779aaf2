"""GPU parity of the bf16x3 dense-layer kernel (csrc/gemm_x3.hip) against float64 references: forward, dX, dW, db,
ragged M/N, all four operand layouts, split-K; and a micro-benchmark line against the fp32 library GEMM."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


@pytest.mark.parametrize("M,K,N,relu", [(41160, 256, 1024, True), (41160, 1024, 256, False), (4000, 256, 768, False),
                                          (1028, 256, 288, False), (31360, 256, 256, False), (516, 2048, 256, True)])
def test_linear_forward_backward_vs_fp64(M, K, N, relu):
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    from combo_avs_amd.ops.linear import linear
    prev_impl = L._IMPL  # (restored below: the implementation switch is process-wide and must not leak into other tests)
    L.set_impl("x3")
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    g = torch.randn(M, N, device="cuda")
    y = linear(x, w, b, relu)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), g)
    xd, wd, bd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    yd = torch.nn.functional.linear(xd, wd, bd)
    if relu:  # use the kernel's own activation pattern: entries with |y| ~ 1e-7 may legitimately differ in sign
        yd = yd * (y.detach() > 0)
    gxd, gwd, gbd = torch.autograd.grad(yd, (xd, wd, bd), g.double())
    # fp32 library GEMM for comparison of the error level
    y32 = torch.nn.functional.linear(x.detach(), w.detach(), b.detach())
    if relu:
        y32 = torch.relu(y32)
    e_x3, e_32 = rel_err(y, yd), rel_err(y32, yd)
    assert e_x3 < 2e-5, (e_x3, e_32)
    assert e_x3 < 50 * e_32 + 1e-6, (e_x3, e_32)  # same class as fp32 round-off (fp32 itself is ~1e-7..1e-6 here)
    L.set_impl(prev_impl)
    assert rel_err(gx, gxd) < 2e-5
    assert rel_err(gw, gwd) < 2e-5
    assert rel_err(gb, gbd) < 2e-5


def test_operand_layouts_and_ragged_edges():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_x3
    torch.manual_seed(0)
    M, N, K = 260, 132, 96
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    ref = (A.double() @ B.double().T)
    for ar in (False, True):
        for br in (False, True):
            a = A.T.contiguous() if ar else A
            b = B.T.contiguous() if br else B
            for splits in (1, 3):
                c = gemm_x3(a, ar, b, br, M, N, K, splits=splits)
                assert rel_err(c, ref) < 2e-5, (ar, br, splits, rel_err(c, ref))


def test_microbench_vs_library(capsys):
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_x3
    M, K, N = 41160, 256, 1024
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")

    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / 20 * 1e3
    t_lib = t(lambda: torch.nn.functional.linear(x, w))
    t_x3 = t(lambda: gemm_x3(x, False, w, False, M, N, K))
    with capsys.disabled():
        print(f"\n[gemm {M}x{K}x{N}] fp32 library {t_lib:.0f} us, bf16x3 MFMA {t_x3:.0f} us ({t_lib / t_x3:.2f}x)")
    assert t_x3 < t_lib


@pytest.mark.parametrize("M,N,K", [(41160, 1024, 256), (41160, 256, 1024), (31360, 256, 256), (41160, 96, 256), (4001, 192, 128),
                                   (4000, 256, 2048), (4000, 2048, 256), (1960, 256, 256), (300, 68, 260), (5000, 3, 256),
                                   (257, 64, 64), (41160, 288, 256)])
def test_weight_gradient_gemm_tn(M, N, K, capsys):
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_tn_x3
    torch.manual_seed(M + N + K)
    dy = torch.randn(M, N, device="cuda")
    x = torch.randn(M, K, device="cuda")
    got, db = gemm_tn_x3(dy, x, with_bias_grad=True)
    ref = dy.double().t() @ x.double()
    assert rel_err(got, ref) < 2e-5, rel_err(got, ref)
    assert rel_err(db, dy.double().sum(0)) < 1e-5

    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / 10 * 1e3
    with capsys.disabled():
        print(f"\n[dW {M}x{N}x{K}] library {t(lambda: dy.t() @ x):.0f} us, gemm_tn_x3 {t(lambda: gemm_tn_x3(dy, x)):.0f} us")


def test_weight_gradient_gemm_tn_strided_views_and_packed_output():
    """Row-strided operands (column blocks of wider tensors) and the reduce writing into a row block of a packed
    gradient (nn.MultiheadAttention's in_proj layout)."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_tn_x3
    torch.manual_seed(3)
    M, N, K = 7840, 256, 256
    big_dy = torch.randn(M, 3 * N, device="cuda")
    big_x = torch.randn(M, K + 64, device="cuda")
    dy, x = big_dy[:, N:2 * N], big_x[:, 64:]
    dW = torch.full((3 * N, K), 7.0, device="cuda")
    db = torch.full((3 * N,), 7.0, device="cuda")
    gemm_tn_x3(dy, x, with_bias_grad=True, out=dW[N:2 * N], db_out=db[N:2 * N])
    ref = dy.double().t() @ x.double()
    assert rel_err(dW[N:2 * N], ref) < 2e-5
    assert rel_err(db[N:2 * N], dy.double().sum(0)) < 1e-5
    assert (dW[:N] == 7).all() and (dW[2 * N:] == 7).all() and (db[:N] == 7).all() and (db[2 * N:] == 7).all()


@pytest.mark.parametrize("M,K,N,bias,relu", [(41160, 256, 1024, True, True), (41160, 1024, 256, True, False),
                                             (41160, 256, 192, True, False), (41160, 256, 96, False, False),
                                             (20001, 64, 288, True, True), (16384, 2048, 256, False, False),
                                             (300, 16, 40, True, False)])
def test_gemm_nt_x3(M, K, N, bias, relu, capsys):
    """C = A B^T (+bias, +ReLU) on csrc/gemm_nt.hip vs fp64; ragged M / N tiles, K a multiple of 16."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_nt_x3
    torch.manual_seed(M + K + N)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda") if bias else None
    got = gemm_nt_x3(a, w, b, relu)
    ref = a.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp_min(0)
    assert rel_err(got, ref) < 2e-5, rel_err(got, ref)

    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / 10 * 1e3

    def lib3():
        torch.backends.cuda.matmul.allow_tf32 = True
        y = torch.nn.functional.linear(a, w, b)
        torch.backends.cuda.matmul.allow_tf32 = False
        return y
    with capsys.disabled():
        print(f"\n[NT {M}x{K}->{N}] hipBLASLt 3xbf16 {t(lib3):.0f} us, gemm_nt_x3 {t(lambda: gemm_nt_x3(a, w, b, relu)):.0f} us")


def test_gemm_nt_x3_strided_token_operand():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_nt_x3
    torch.manual_seed(5)
    big = torch.randn(20000, 512, device="cuda")
    a = big[:, 128:384]  # row stride 512, 16-byte aligned start
    w = torch.randn(200, 256, device="cuda")
    assert rel_err(gemm_nt_x3(a, w), a.double() @ w.double().t()) < 2e-5


@pytest.mark.parametrize("M,K,N", [(41160, 256, 1024), (20001, 64, 288), (777, 2048, 130), (16400, 16, 40)])
def test_gemm_nt_v2_is_bitwise_v1_and_accepts_strided_weight_views(M, K, N):
    """csrc/gemm_nt2.hip (persistent tiles, pre-split weight image) must reproduce csrc/gemm_nt.hip bit for bit - the same
    products in the same order - on full, ragged and tiny tiles, for W and for a W^T view (dX = dY . W without a
    transpose copy), with and without bias / ReLU."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    torch.manual_seed(M)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")
    b = torch.randn(N, device="cuda")
    prev = L.NT_V2
    try:
        L.NT_V2 = False
        y1 = L.gemm_nt_x3(a, w, b, relu=True)
        z1 = L.gemm_nt_x3(a, w.t().contiguous().t())
        L.NT_V2 = True
        y2 = L.gemm_nt_x3(a, w, b, relu=True)
        z2 = L.gemm_nt_x3(a, w.t().contiguous().t())  # a [N, K] view with strides (1, N)
    finally:
        L.NT_V2 = prev
    assert torch.equal(y1, y2)
    assert torch.equal(z1, z2)
    ref = torch.relu(a.double() @ w.double().t() + b.double())
    assert rel_err(y2, ref) < 2e-5


def test_mask_logit_contraction_and_gradients_on_the_hip_gemms():
    """ops/masklogit.py: `einsum("bqc,bchw->bqhw")` per prediction head on the batched gemm_nt2 kernel (128 x 128 tiles for
    the 100 queries of a frame, mask features pre-split once) and its two gradients (batched NT GEMM against the
    transposed image; grouped TN GEMM, one problem per frame) against float64 and against the library path."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import masklogit as ML
    torch.manual_seed(3)
    bt, Q, HW, C, nh = 3, 100, 56 * 56, 256, 4
    mf = torch.randn(bt, HW, C, device="cuda", requires_grad=True)
    mes = [torch.randn(bt, Q, C, device="cuda", requires_grad=True) for _ in range(nh)]
    g = torch.randn(nh, bt, Q, HW, device="cuda")

    def run(hip):
        prev, prev_f = ML.HIP_GEMMS, ML.HIP_FORWARD
        ML.HIP_GEMMS = ML.HIP_FORWARD = hip  # (the product default keeps the forward on the library, see ops/masklogit.py)
        try:
            buf = torch.empty(nh, bt, Q, HW, device="cuda")
            img = ML.prepare(mf)
            assert (img is not None) == hip
            for i in range(nh):
                ML.mask_logits_into(mes[i], mf, buf[i], img)
            out = ML.attach_mask_logit_grads(mf, buf, mes)
            grads = torch.autograd.grad(out, [mf] + mes, g)
            return buf, grads
        finally:
            ML.HIP_GEMMS, ML.HIP_FORWARD = prev, prev_f
    buf_h, gr_h = run(True)
    buf_l, gr_l = run(False)
    ref = torch.stack([m.detach().double() @ mf.detach().double().transpose(1, 2) for m in mes])
    assert rel_err(buf_h, ref) < 2e-5, rel_err(buf_h, ref)
    gd = g.double()
    ref_dmf = sum(gd[i].transpose(1, 2) @ mes[i].detach().double() for i in range(nh))
    assert rel_err(gr_h[0], ref_dmf) < 2e-5, rel_err(gr_h[0], ref_dmf)
    for i in range(nh):
        assert rel_err(gr_h[1 + i], gd[i] @ mf.detach().double()) < 2e-5
    for a, b in zip(gr_h, gr_l):
        assert rel_err(a, b.double()) < 2e-5


@pytest.mark.parametrize("M,C,Hd", [(41160, 256, 1024), (4000, 256, 2048), (700, 256, 512)])
def test_ffn_relu_gradient_folded_into_the_dx_gemm(M, C, Hd):
    """ops.linear.ffn: linear2(relu(linear1(x))) with the ReLU backward applied in the epilogue of linear2's input-gradient
    GEMM (csrc/gemm_nt2.hip, mask operand) must give the gradients of the unfused chain bit for bit where both take the
    HIP path, and float64's to 2e-5 everywhere (small shapes fall back to the library + the ReLU-gradient kernel)."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    torch.manual_seed(C + Hd)
    x = torch.randn(M, C, device="cuda", requires_grad=True)
    w1 = (torch.randn(Hd, C, device="cuda") / C ** 0.5).requires_grad_(True)
    b1 = torch.randn(Hd, device="cuda", requires_grad=True)
    w2 = (torch.randn(C, Hd, device="cuda") / Hd ** 0.5).requires_grad_(True)
    b2 = torch.randn(C, device="cuda", requires_grad=True)
    g = torch.randn(M, C, device="cuda")

    def run(fused):
        prev, L.FFN_FUSED_RELU_GRAD = L.FFN_FUSED_RELU_GRAD, fused
        try:
            y = L.ffn(x, w1, b1, w2, b2, defer=False)
            return y.detach(), torch.autograd.grad(y, (x, w1, b1, w2, b2), g)
        finally:
            L.FFN_FUSED_RELU_GRAD = prev
    y_f, g_f = run(True)
    y_u, g_u = run(False)
    assert torch.equal(y_f, y_u)
    for a, b in zip(g_f, g_u):
        assert rel_err(a, b) < 1e-6, rel_err(a, b)
    xd, w1d, b1d, w2d, b2d = (t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    hd = torch.nn.functional.linear(xd, w1d, b1d)
    # the product's own activation pattern: hidden entries with |h| ~ 1e-6 may legitimately differ in sign from fp64
    hd = hd * (L.linear(x.detach(), w1.detach(), b1.detach(), relu=True) > 0)
    yd = torch.nn.functional.linear(hd, w2d, b2d)
    gd = torch.autograd.grad(yd, (xd, w1d, b1d, w2d, b2d), g.double())
    assert rel_err(y_f, yd) < 5e-5
    for a, b in zip(g_f, gd):
        assert rel_err(a, b) < 5e-5, rel_err(a, b)
