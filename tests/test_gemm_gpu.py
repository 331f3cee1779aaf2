"""GPU parity of the head's dense-layer kernels against float64 references (the fp16-piece forward mode has its own file,
tests/test_f16x3_gpu.py; `ops.linear.linear` below runs in the process's default forward mode): the exact-fp32 MFMA forward GEMM
(csrc/gemm_f32.hip; must sit in fp32's own error class, not the bf16 split's), the 3-product bf16 input-gradient GEMM
(csrc/gemm_nt3.hip) and weight-gradient GEMM (csrc/gemm_tn.hip): ragged M/N, strided operands, batched form, deferred /
grouped weight gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def _t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


@pytest.mark.parametrize("M,K,N,relu", [(41160, 256, 1024, True), (41160, 1024, 256, False), (4000, 256, 768, False),
                                          (1028, 256, 288, False), (31360, 256, 256, False), (516, 2048, 256, True),
                                          (40, 128, 4096, True), (4000, 256, 3, False)])
def test_linear_forward_backward_vs_fp64(M, K, N, relu):
    """ops.linear.linear: forward at fp32 grade - the default fp16-piece products or, with COMBO_HEAD_FORWARD=fp32, the exact-fp32 MFMA
    kernel (error of fp32's own class either way: compared with the library's
    fp32 GEMM on the same data), dX / dW / db on the 3-product bf16 kernels (2e-5)."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import linear
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
    y32 = torch.nn.functional.linear(x.detach(), w.detach(), b.detach())  # fp32 library GEMM: the error class to match
    if relu:
        y32 = torch.relu(y32)
    e_own, e_32 = rel_err(y, yd), rel_err(y32, yd)
    assert e_own < 1e-6, (e_own, e_32)
    assert e_own < 3 * e_32 + 1e-7, (e_own, e_32)
    assert rel_err(gx, gxd) < 2e-5
    assert rel_err(gw, gwd) < 2e-5
    assert rel_err(gb, gbd) < 2e-5


@pytest.mark.parametrize("M,K,N,bias,relu", [(41160, 256, 1024, True, True), (41160, 1024, 256, True, False),
                                             (41160, 256, 288, True, False), (41160, 256, 96, False, False),
                                             (20001, 64, 288, True, True), (16384, 2048, 256, False, False),
                                             (300, 16, 40, True, False), (4000, 256, 256, True, False),
                                             (4000, 2048, 256, True, False), (1960, 256, 512, True, False),
                                             (100, 256, 3136, False, False), (3, 16, 5, True, False),
                                             (1960, 2048, 256, True, True), (7840, 1024, 256, False, False)])  # split-K shapes
def test_gemm_nt_f32_exact(M, K, N, bias, relu, capsys):
    """C = A B^T (+bias, +ReLU) on csrc/gemm_f32.hip vs fp64: ragged M / N tiles, every tile configuration (COMBO_F32_TILE
    is not set: the launcher's own choice), error of fp32 round-off (~1e-7), never the bf16 split's 4e-6."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_nt_f32
    torch.manual_seed(M + K + N)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda") if bias else None
    got = gemm_nt_f32(a, w, b, relu)
    ref = a.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp_min(0)
    e = rel_err(got, ref)
    assert e < 6e-7 * max(1.0, (K / 256) ** 0.5), e  # fp32 accumulation: round-off grows ~sqrt(K)
    if M * N * K >= 1e9:
        us = _t(lambda: gemm_nt_f32(a, w, b, relu))
        with capsys.disabled():
            print(f"\n[f32 NT {M}x{K}->{N}] gemm_nt_f32 {us:.0f} us = {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s of 157.3 fp32-MFMA peak; "
                  f"library fp32 {_t(lambda: torch.nn.functional.linear(a, w, b)):.0f} us")


def test_gemm_nt_f32_strided_operands_column_block_output_and_batched_form():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    from combo_avs_amd.ops.linear import gemm_nt_f32
    torch.manual_seed(5)
    big = torch.randn(20000, 512, device="cuda")
    a = big[:, 128:384]  # row stride 512, 16-byte aligned start
    wbig = torch.randn(768, 256, device="cuda")
    w = wbig[256:512]  # a row block of a packed weight (nn.MultiheadAttention's in_proj)
    out = torch.full((20000, 520), 7.0, device="cuda")
    gemm_nt_f32(a, w, out=out[:, 256:512])
    assert rel_err(out[:, 256:512], a.double() @ w.double().t()) < 6e-7
    assert (out[:, :256] == 7).all() and (out[:, 512:] == 7).all()
    # batched: the mask-logit contraction, one problem per frame
    bt, Q, HW, C = 3, 100, 56 * 56, 256
    me, mf = torch.randn(bt, Q, C, device="cuda"), torch.randn(bt, HW, C, device="cuda")
    o = torch.full((bt + 1, Q, HW), 7.0, device="cuda")
    _lib.check(_lib.lib().combo_gemm_nt_batched_f32(me.data_ptr(), C, Q * C, mf.data_ptr(), C, HW * C, o.data_ptr(), HW, Q * HW,
                                                    Q, HW, C, bt, 0, _lib.current_stream()), "combo_gemm_nt_batched_f32")
    assert rel_err(o[:bt], me.double() @ mf.double().transpose(1, 2)) < 6e-7
    assert (o[bt] == 7).all()


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
def test_gemm_nt_f32_every_tile_configuration_agrees(tile, monkeypatch):
    """wide / mid / skinny / tall tiles forced through the C ABI's A/B switch in a child process (the switch is read once)."""
    import os
    import subprocess
    import sys
    code = (
        "import torch, combo_avs_amd\n"
        "from combo_avs_amd.ops.linear import gemm_nt_f32\n"
        "torch.manual_seed(1)\n"
        "for M, K, N in ((1000, 256, 320), (257, 2048, 70), (33, 16, 129)):\n"
        "    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')\n"
        "    got = gemm_nt_f32(a, w, b, True)\n"
        "    ref = torch.relu(a.double() @ w.double().t() + b.double())\n"
        "    e = ((got.double() - ref).norm() / ref.norm()).item()\n"
        "    assert e < 6e-7 * max(1.0, (K / 256) ** 0.5), (M, K, N, e)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, COMBO_F32_TILE=str(tile), PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]

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
    with capsys.disabled():
        print(f"\n[dW {M}x{N}x{K}] library {_t(lambda: dy.t() @ x):.0f} us, gemm_tn_x3 {_t(lambda: gemm_tn_x3(dy, x)):.0f} us")


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


@pytest.mark.parametrize("M,K,N,relu_mask", [(41160, 1024, 256, False), (41160, 256, 1024, True), (41160, 288, 256, False),
                                               (20001, 64, 288, False), (16384, 2048, 256, False), (300, 16, 40, False),
                                               (4000, 256, 256, False), (4000, 256, 2048, True), (4000, 512, 256, False)])
def test_gemm_nt_x3_input_gradient(M, K, N, relu_mask):
    """dX = dY . W on csrc/gemm_nt3.hip: the weight arrives as a W^T VIEW (pre-split without a transpose copy), optional
    ReLU-mask epilogue; 3-product bf16 split with a rounded `hi` part: < 1e-5 against fp64."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_nt_x3
    torch.manual_seed(M + K + N)
    dy = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda") * 0.1  # nn.Linear weight [out = K of this GEMM, in = N]
    mask = torch.randn(M, N, device="cuda") if relu_mask else None
    got = gemm_nt_x3(dy, w.t(), relu_mask=mask)
    ref = dy.double() @ w.double()
    if relu_mask:
        ref = ref * (mask > 0)
    assert rel_err(got, ref) < 1e-5, rel_err(got, ref)


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
@pytest.mark.parametrize("M,K,N", [(5000, 256, 384), (777, 48, 200), (20001, 64, 288)])
def test_gemm_nt_x3_every_tile_configuration_agrees(tile, M, K, N):
    """csrc/gemm_nt3.hip: wide / mid / skinny / tall tiles forced (combo_gemm_nt_x3_tile) - ragged M and N, an odd number of K stages
    (K = 48: the register ping-pong hands its sets over at the tile boundary), bias + ReLU and the masked epilogue"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    from combo_avs_amd.ops.linear import gemm_nt_x3
    torch.manual_seed(tile * 1000 + M)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda")
    mask = torch.randn(M, N, device="cuda")
    lib = _lib.lib()
    prev = lib.combo_gemm_nt_x3_tile(tile)
    try:
        got = gemm_nt_x3(a, w, bias=b, relu=True)
        got_m = gemm_nt_x3(a, w, relu_mask=mask)
        again = gemm_nt_x3(a, w, bias=b, relu=True)
    finally:
        lib.combo_gemm_nt_x3_tile(prev)
    ref = a.double() @ w.double().t()
    assert rel_err(got, torch.relu(ref + b.double())) < 1e-5
    assert rel_err(got_m, ref * (mask > 0)) < 1e-5
    assert torch.equal(got, again)  # deterministic


@pytest.mark.parametrize("M,K,N", [(1, 16, 4), (33, 32, 37), (257, 64, 130), (4000, 2048, 256), (1960, 512, 2048)])
def test_gemm_nt_x3_edges_scalar_stores_single_row_and_split_k(M, K, N):
    """csrc/gemm_nt3.hip beyond the main road: one row; N not a multiple of 4 (scalar stores, scalar mask loads); shapes the
    split-K plan takes (few output tiles, long K: partial sums + the fixed-order finishing launch, with and without the mask)"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    from combo_avs_amd.ops.linear import gemm_nt_x3
    torch.manual_seed(M + N)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.1
    mask = torch.randn(M, N, device="cuda")
    ref = a.double() @ w.double().t()
    got, got_m, again = gemm_nt_x3(a, w), gemm_nt_x3(a, w, relu_mask=mask), gemm_nt_x3(a, w)
    # (relative to the UNMASKED result's norm: the one-row case keeps two small values, 2^-17 of the products they came from)
    assert rel_err(got, ref) < 1e-5 and float((got_m.double() - ref * (mask > 0)).norm() / ref.norm()) < 1e-5
    assert torch.equal(got, again)
    if (M, K, N) in ((4000, 2048, 256), (1960, 512, 2048)):
        assert (_lib.lib().combo_gemm_nt_x3_splitk_plan(M, N, K) > 1) == ((M, K, N) == (4000, 2048, 256))


def test_gemm_nt_x3_strided_token_operand():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_nt_x3
    torch.manual_seed(5)
    big = torch.randn(20000, 512, device="cuda")
    a = big[:, 128:384]  # row stride 512, 16-byte aligned start
    w = torch.randn(200, 256, device="cuda")
    assert rel_err(gemm_nt_x3(a, w), a.double() @ w.double().t()) < 1e-5


def test_mask_logit_contraction_and_gradients():
    """ops/masklogit.py: `einsum("bqc,bchw->bqhw")` per prediction head on the batched exact-fp32 kernel and its two
    gradients (batched NT GEMM against the transposed image; grouped TN GEMM, one problem per frame) against float64."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import masklogit as ML
    torch.manual_seed(3)
    bt, Q, HW, C, nh = 3, 100, 56 * 56, 256, 4
    mf = torch.randn(bt, HW, C, device="cuda", requires_grad=True)
    mes = [torch.randn(bt, Q, C, device="cuda", requires_grad=True) for _ in range(nh)]
    g = torch.randn(nh, bt, Q, HW, device="cuda")
    buf = torch.empty(nh, bt, Q, HW, device="cuda")
    for i in range(nh):
        ML.mask_logits_into(mes[i], mf, buf[i])
    out = ML.attach_mask_logit_grads(mf, buf, mes)
    grads = torch.autograd.grad(out, [mf] + mes, g)
    ref = torch.stack([m.detach().double() @ mf.detach().double().transpose(1, 2) for m in mes])
    assert rel_err(buf, ref) < 6e-7, rel_err(buf, ref)
    gd = g.double()
    ref_dmf = sum(gd[i].transpose(1, 2) @ mes[i].detach().double() for i in range(nh))
    assert rel_err(grads[0], ref_dmf) < 2e-5, rel_err(grads[0], ref_dmf)
    for i in range(nh):
        assert rel_err(grads[1 + i], gd[i] @ mf.detach().double()) < 2e-5


def test_deferred_weight_gradient_with_a_non_deferrable_use_of_the_same_weight():
    """one weight used twice inside deferred_dw(), once with enough rows for the grouped kernel and once with too
    few (M < 256).  The second use must neither be lost nor be added to the not-yet-written destination."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import linear as L
    torch.manual_seed(11)
    K, N = 256, 128
    w = (torch.randn(N, K, device="cuda") / 16).requires_grad_(True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    xa = torch.randn(2000, K, device="cuda")
    xb = torch.randn(40, K, device="cuda")
    for order in ((xa, xb), (xb, xa)):
        with L.deferred_dw():  # the trainer's pattern: autograd.grad inside the context, results read after it closed
            y = sum(L.linear(x, w, b, defer=True).square().sum() for x in order)
            gw, gb = torch.autograd.grad(y, (w, b))
        ref_w, ref_b = torch.autograd.grad(sum(torch.nn.functional.linear(x.double(), w.double(), b.double()).square().sum()
                                               for x in order), (w, b))
        assert rel_err(gw, ref_w) < 2e-5, rel_err(gw, ref_w)
        assert rel_err(gb, ref_b) < 2e-5


@pytest.mark.parametrize("M,C,Hd", [(41160, 256, 1024), (4000, 256, 2048), (700, 256, 512)])
def test_ffn_relu_gradient_folded_into_the_dx_gemm(M, C, Hd):
    """ops.linear.ffn: linear2(relu(linear1(x))) with the ReLU backward applied in the epilogue of linear2's input-gradient
    GEMM (csrc/gemm_nt3.hip, mask operand) must give the gradients of the unfused chain bit for bit where both take the
    HIP path, and float64's to 2e-5 everywhere ."""
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


@pytest.mark.parametrize("M,K,N,relu", [(40, 128, 4096, True), (40, 4096, 4096, True), (40, 4096, 256, False), (5, 4096, 4096, True),
                                          (64, 256, 1024, False), (17, 192, 1000, True)])
def test_gemm_smallm_weight_streaming(M, K, N, relu, capsys):
    """csrc/gemm_smallm.hip (audio_mlp, audio_transformation.py:13-14): a few rows against a large weight, exact fp32, with and
    without K splits, ragged N; timing of the 67 MB layer against its HBM time."""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.linear import gemm_smallm_f32
    torch.manual_seed(M + K + N)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    got = gemm_smallm_f32(a, w, b, relu)
    ref = a.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    assert rel_err(got, ref) < 6e-7 * max(1.0, (K / 256) ** 0.5), rel_err(got, ref)
    if K * N >= 1 << 24:
        us = _t(lambda: gemm_smallm_f32(a, w, b, relu), 20)
        with capsys.disabled():
            print(f"\n[smallm {M}x{K}->{N}] {us:.1f} us = {K * N * 4 / us / 1e6:.2f} TB/s of weights "
                  f"(library: {_t(lambda: torch.nn.functional.linear(a, w, b), 20):.1f} us)")


def test_grouped_weight_presplit_gives_bitwise_the_same_input_gradients():
    """Inside grouped_presplit() the images of all announced weights are split by one grouped launch (ops.linear.expect_input_grad,
    csrc/gemm_x3.hip presplit_grouped_kernel): packed q/k/v slices, a column-concatenated pair and plain layers, compared
    with the per-weight route outside the context - same split, same GEMM, so bit-identical."""
    from combo_avs_amd.ops import linear as L
    torch.manual_seed(3)
    dev, E = "cuda", 256
    x = torch.randn(4000, E, device=dev, requires_grad=True)
    mem = torch.randn(6272, E, device=dev, requires_grad=True)
    W = torch.randn(3 * E, E, device=dev, requires_grad=True) * 0.05
    b = torch.randn(3 * E, device=dev)
    w1, b1 = torch.randn(192, E, device=dev) * 0.05, torch.randn(192, device=dev)
    w2, b2 = torch.randn(96, E, device=dev) * 0.05, torch.randn(96, device=dev)
    wf1, bf1 = torch.randn(1024, E, device=dev) * 0.05, torch.randn(1024, device=dev)
    ws, bs = (torch.randn(E, 32, device=dev) * 0.05).requires_grad_(True), torch.randn(E, device=dev)  # 32 x 256 image: < one workgroup
    xs = torch.randn(4000, 32, device=dev, requires_grad=True)
    wf2, bf2 = torch.randn(E, 1024, device=dev) * 0.05, torch.randn(E, device=dev)
    for t in (w1, w2, wf1, wf2):
        t.requires_grad_(True)

    def run():
        q, k, v = L.in_proj(x, x, x, W, b, same_qk=True, defer=True)
        q2, k2, v2 = L.in_proj(x, mem, mem, W, b, same_qk=False, defer=True)
        c = L.linear_cat(x, w1, b1, w2, b2, defer=True)
        f = L.ffn(x, wf1, bf1, wf2, bf2)
        sm = L.linear(xs, ws, bs, defer=True)  # a weight whose image is smaller than one workgroup of the grouped split
        loss = (q * k).sum() + v.square().sum() + q2.sum() * 0.3 + (k2 * v2).sum() + c.square().sum() + f.square().sum() + sm.square().sum()
        return torch.autograd.grad(loss, [x, mem, xs])

    ref = run()
    with L.grouped_presplit(), L.deferred_dw():
        got = run()
        assert not L._split_pending and len(L._split_images) == 8  # W[:2E], W[:E], W[E:2E], W[2E:], (w1,w2), wf1, wf2, ws
    assert L._split_images is None
    for r, g in zip(ref, got):
        assert torch.equal(r, g)


@pytest.mark.parametrize("M,K,N", [(3000, 512, 256), (777, 128, 52), (41160, 64, 256)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_gemm_nt_x3_epilogue_with_add_and_mask_together(M, K, N, tile):
    """combo_gemm_nt_x3_epi2_f32: bias + add + ReLU + mask in one epilogue (the add and mask tensors share one register set in the
    kernel), every tile configuration, with and without split-K, N not a multiple of 4 (scalar stores)"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    from combo_avs_amd.ops.linear import presplit
    torch.manual_seed(M + N + tile)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    add = torch.randn(M, N, device="cuda")
    mask = torch.randn(M, N, device="cuda")
    img = presplit(w)
    ref = (torch.relu(a.double() @ w.double().t() + b.double() + add.double())) * (mask.double() > 0)
    lib, st = _lib.lib(), _lib.current_stream()
    prev = lib.combo_gemm_nt_x3_tile(tile)
    try:
        for splits in ((1, 2, 4) if (N % 4 == 0 and K % 128 == 0 and K >= 512) else (1,)):
            y = torch.full((M, N), float("nan"), device="cuda")
            ws = torch.empty(splits, M, N, device="cuda") if splits > 1 else None
            rc = lib.combo_gemm_nt_x3_epi2_f32(a.data_ptr(), K, img.data_ptr(), b.data_ptr(), add.data_ptr(), mask.data_ptr(), y.data_ptr(), N,
                                               M, N, K, 1, splits, _lib.ptr(ws), st)
            assert rc == 0, (rc, splits)
            assert rel_err(y, ref) < 1e-5, (splits, rel_err(y, ref))
    finally:
        lib.combo_gemm_nt_x3_tile(prev)
