"""GPU parity of the merged K / V projections of a memory level (ops/linear.py memory_kv + in_proj_q + the strided dK / dV
outputs of csrc/attention.hip) against the per-layer packed projection (ops/linear.py in_proj), which the golden-fixture tests
pinned to nn.MultiheadAttention (transformer_decoder.py:99-118 of the reference)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

E, H = 256, 8


def _problem(B, hw, Q, layers, seed=0):
    g = torch.Generator().manual_seed(seed)

    def r(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).cuda()
    mem = r(B, E, hw).transpose(1, 2)  # the decoder's memory is a transposed [B, C, hw] view
    pos = r(1, hw, E, scale=0.5)
    qs = [r(B, Q, E) for _ in range(layers)]
    Ws = [r(3 * E, E, scale=E ** -0.5) for _ in range(layers)]
    bs = [r(3 * E, scale=0.1) for _ in range(layers)]
    probes = [r(B * Q, E) for _ in range(layers)]
    blocked = (torch.rand(B, Q, (hw + 3) // 4 * 4, generator=g) > 0.7).to(torch.uint8).cuda()
    blocked[:, :, 0] = 0  # no fully blocked row
    return mem, pos, qs, Ws, bs, probes, blocked


def _leaves(mem, pos, qs, Ws, bs):
    return (mem.detach().clone().requires_grad_(True), pos.detach().clone().requires_grad_(True),
            [q.detach().clone().requires_grad_(True) for q in qs], [w.detach().clone().requires_grad_(True) for w in Ws],
            [b.detach().clone().requires_grad_(True) for b in bs])


def _per_layer(mem, pos, qs, Ws, bs, probes, blocked, used):
    from combo_avs_amd.ops.attention import attention
    from combo_avs_amd.ops.linear import in_proj
    B, hw, Q = mem.shape[0], mem.shape[1], qs[0].shape[1]
    mem_k = mem + pos
    outs = []
    for j, (q_in, W, b) in enumerate(zip(qs, Ws, bs)):
        q, k, v = in_proj(q_in, mem_k, mem, W, b, defer=True)
        outs.append(attention(q.reshape(B * Q, E), k.reshape(B * hw, E), v.reshape(B * hw, E), blocked, B, H))
    return outs, sum((o * p).sum() for j, (o, p) in enumerate(zip(outs, probes)) if j in used)


def _merged(mem, pos, qs, Ws, bs, probes, blocked, used):
    from combo_avs_amd.ops.attention import attention
    from combo_avs_amd.ops.linear import in_proj_q, memory_kv
    B, Q = mem.shape[0], qs[0].shape[1]
    kv = memory_kv([mem + pos], [mem], [list(zip(Ws, bs))], defer=True)[0]
    outs = []
    for j, (q_in, W, b) in enumerate(zip(qs, Ws, bs)):
        q = in_proj_q(q_in, W, b, defer=True)
        outs.append(attention(q.reshape(B * Q, E), kv[j][0], kv[j][1], blocked, B, H))
    return outs, sum((o * p).sum() for j, (o, p) in enumerate(zip(outs, probes)) if j in used)


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("mode", ["eager", "deferred", "deferred_flat"])
@pytest.mark.parametrize("B,hw,Q,layers", [(4, 196, 24, 3), (2, 787, 100, 3), (3, 49, 7, 2)])
def test_memory_kv_equals_the_per_layer_projection(mode, B, hw, Q, layers):
    """forward outputs and every gradient (memory, position, queries, the packed weights and biases) of the merged path equal
    the per-layer path; `deferred`: inside deferred_dw() + grouped_presplit(), where the q node and the level's k / v node fill
    ONE packed gradient tensor; `deferred_flat`: that tensor is a registered flat-buffer view (trainer.FlatAdamW)"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import attention as A
    from combo_avs_amd.ops import linear as L
    base = _problem(B, hw, Q, layers, seed=B * 1000 + hw)
    probes, blocked = base[5], base[6]
    used = set(range(layers))
    A.kv_gradient_slots.clear()  # (forward passes of earlier tests that never ran backward leave dead weak references)
    got_inplace = []
    orig = L._KVGrads.blocks

    def counting(self, j):
        got_inplace.append(j)
        return orig(self, j)
    results = []
    for fn in (_per_layer, _merged):
        mem, pos, qs, Ws, bs = _leaves(*base[:5])
        leaves = [mem, pos] + qs + Ws + bs
        flat = torch.full((sum(w.numel() for w in Ws),), float("nan"), device="cuda")
        views, off = {}, 0
        for w in Ws:
            views[w.data_ptr()] = flat[off:off + w.numel()].view_as(w)
            off += w.numel()
        L._KVGrads.blocks = counting
        try:
            if mode == "eager":
                outs, loss = fn(mem, pos, qs, Ws, bs, probes, blocked, used)
                grads = torch.autograd.grad(loss, leaves)
            else:
                with L.grouped_presplit():
                    outs, loss = fn(mem, pos, qs, Ws, bs, probes, blocked, used)
                    with L.grad_targets(views if mode == "deferred_flat" else {}), L.deferred_dw():
                        grads = torch.autograd.grad(loss, leaves)
        finally:
            L._KVGrads.blocks = orig
        if mode == "deferred_flat" and fn is _merged:
            for w, g in zip(Ws, grads[2 + layers:2 + 2 * layers]):
                assert g.data_ptr() == views[w.data_ptr()].data_ptr(), "packed gradient was not written into the flat-buffer view"
        results.append(([o.clone() for o in outs], [g.clone() for g in grads]))
    assert sorted(got_inplace) == list(range(layers)), "dk / dv were not written in place by the strided attention backward"
    assert not A.kv_gradient_slots, "gradient slots must be released by the backward pass"
    assert not L._packed_grads
    (o_ref, g_ref), (o_got, g_got) = results
    for a, b in zip(o_got, o_ref):
        assert _rel(a, b) < 2e-6
    names = ["memory", "pos"] + [f"query{j}" for j in range(layers)] + [f"W{j}" for j in range(layers)] + [f"b{j}" for j in range(layers)]
    for n, a, b in zip(names, g_got, g_ref):
        assert a.shape == b.shape and torch.isfinite(a).all(), n
        assert _rel(a, b) < 3e-5, (n, _rel(a, b))  # 3-product bf16 split GEMMs, different K grouping (K = 768 vs 3 x 256)


def test_memory_kv_layer_without_loss_gets_zero_rows():
    """a layer whose attention output reaches no loss: its k / v gradient blocks are zero (memory gradient = the other layers'),
    its packed weight gradient has zero k / v rows and - nobody wrote them - zero q rows"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops import attention as A
    from combo_avs_amd.ops import linear as L
    base = _problem(2, 196, 24, 3, seed=5)
    probes, blocked = base[5], base[6]
    out = []
    for fn in (_per_layer, _merged):
        mem, pos, qs, Ws, bs = _leaves(*base[:5])
        with L.grouped_presplit():
            _, loss = fn(mem, pos, qs, Ws, bs, probes, blocked, {0, 2})
            with L.deferred_dw():
                out.append(torch.autograd.grad(loss, [mem, pos, Ws[0], Ws[1], Ws[2], bs[1]], allow_unused=True))
    A.kv_gradient_slots.clear()
    ref, got = out
    for i, (a, b) in enumerate(zip(got, ref)):
        if b is None:  # the per-layer path never reaches layer 1's weight
            assert a is None or float(a.abs().max()) == 0.0, i
        else:
            assert _rel(a, b) < 3e-5, (i, _rel(a, b))


def test_attention_backward_strided_outputs_are_bit_identical():
    """combo_attention_backward_ld_f32 with a 3 x E row pitch writes the same dk / dv bits as the dense entry point"""
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import _lib
    B, Lq, Lk = 3, 50, 333
    g = torch.Generator().manual_seed(3)
    q, k, v, do = [torch.randn(n, E, generator=g).cuda() for n in (B * Lq, B * Lk, B * Lk, B * Lq)]
    out = torch.empty(B * Lq, E, device="cuda")
    lse = torch.empty(B, H, Lq, device="cuda")
    lib, st = _lib.lib(), _lib.current_stream()
    scale = 32 ** -0.5
    _lib.check(lib.combo_attention_forward_f32(q.data_ptr(), E, k.data_ptr(), E, v.data_ptr(), E, 0, 0, 0, 0, B, H, Lq, Lk, scale,
                                               out.data_ptr(), lse.data_ptr(), st), "fwd")
    dq0, dk0, dv0 = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Lq, device="cuda")
    _lib.check(lib.combo_attention_backward_f32(q.data_ptr(), E, k.data_ptr(), E, v.data_ptr(), E, 0, 0, 0, 0, B, H, Lq, Lk, scale,
                                                out.data_ptr(), lse.data_ptr(), do.data_ptr(), delta.data_ptr(), dq0.data_ptr(),
                                                dk0.data_ptr(), dv0.data_ptr(), st), "bwd")
    dq1 = torch.empty_like(q)
    dK = torch.full((B * Lk, 3 * E), 7.0, device="cuda")
    dV = torch.full((B * Lk, 3 * E), 9.0, device="cuda")
    _lib.check(lib.combo_attention_backward_ld_f32(q.data_ptr(), E, k.data_ptr(), E, v.data_ptr(), E, 0, 0, 0, 0, B, H, Lq, Lk, scale,
                                                   out.data_ptr(), lse.data_ptr(), do.data_ptr(), delta.data_ptr(), dq1.data_ptr(),
                                                   dK[:, E:].data_ptr(), 3 * E, dV[:, 2 * E:].data_ptr(), 3 * E, st), "bwd_ld")
    assert torch.equal(dq0, dq1) and torch.equal(dK[:, E:2 * E], dk0) and torch.equal(dV[:, 2 * E:], dv0)
    assert float(dK[:, :E].min()) == 7.0 and float(dK[:, 2 * E:].max()) == 7.0 and float(dV[:, :2 * E].min()) == 9.0
