"""The captured-hipGraph training step (trainer.GraphedTrainStep) against the eager step on the same state and inputs:
same 39 losses, same gradients, same parameters after the update; a second batch goes through the static input
buffers; fused-fusion dropout masks change from replay to replay (device-side step counter)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import graph_compare as GC  # noqa: E402


@pytest.fixture(scope="module")
def rig():
    return GC.build("r50")


# Round 6: the PVT recipe's bound is the R50 recipe's again (head 5e-3 / 25 %, backbones 2e-2).  Round 5 had widened it to 2e-1 for "one run
# in twelve in which every parameter of the last decoder layer moved by 1.1e-1".  Root cause (tools/pvt_forward_states.py,
# tools/graph_outlier_probe.py, profiles/r06_pvt_eager_vs_replay_states.txt): MIOpen's DEFAULT (immediate-mode) solver for the bf16
# spatial-reduction convolutions of PVTv2 (`Attention.sr`, kernel = stride, pvtv2.py:76) is not run-to-run reproducible - the
# convolution's output is the first tensor of a forward pass that differs bitwise between two runs on identical inputs (6 % of the
# passes; block2.2's 128 -> 128, k = 4 layer), bf16 re-rounding in the 40 blocks behind it turns that into a different
# forward pass (losses 2e-4 apart), and eager / replayed steps fall into 2 - 4 reproducible clusters 2 - 6e-3 apart over the head's
# parameters (up to 5e-2 .. 1.1e-1 on single tensors) - between two EAGER steps as well, with the package's own bf16 kernels (SRA
# attention, pre-norm, deferred column sums) on or off, and never with fp32 backbones or the R50 recipe.  With deterministic
# library solvers (`torch.backends.cudnn.deterministic = True`, PyTorch's own switch) 4 eager steps, 16 replays of 8 separate
# captures and one more eager step agree to 1.4e-6 over the head's parameters (backbones: <= 5.3e-3, the library's bf16 weight-
# gradient GEMMs); with MIOpen's find mode (`cudnn.benchmark`, what bench.py runs) to 4.6e-4.  The test below therefore runs under
# `torch.backends.cudnn.flags(deterministic=True)` and holds the head to the strict bound.
PVT_LIB_FRAC = 1.0  # bf16 backbones: relative L2 <= 2e-2 per parameter; the 2e-3 element-wise fraction is not applied to them


def _check_pair(i, ref_l, got_l, report, ref_p, got_p, rel_l2=GC.REL_L2, frac=GC.FRAC, lib_frac=None, param_frac=None):
    assert len(got_l) == 39
    for k, v in got_l.items():
        assert abs(v - ref_l[k]) <= (2e-4 if rel_l2 == GC.REL_L2 else 5e-3) * abs(ref_l[k]) + 1e-5, (i, k, v, ref_l[k])
    # EVERY parameter of the optimiser's table on its own (round 4): relative L2 error <= 5e-3 and <= 25 % of ITS entries beyond
    # 2e-3 RMS + 2e-3 rel (R50 recipe; the noise between an eager and a replayed step is the library's weight-gradient atomics:
    # measured <= 2.3e-3 / 16 % on pre_sam_backbone.stem.conv1.weight, median 5e-5).  A whole-buffer fraction cannot see the
    # biases / level embeddings / norm affines (together far below 1 % of the 87 M entries) - which is how the wrong memset-node
    # gradients of rounds 1-2 (relative error ~1) got through.
    bad = GC.failures(report, rel_l2, frac, lib_frac)
    worst = sorted(report, key=lambda r: -r[3])[:5]
    print(f"[graph vs eager, batch {i}] {len(report)} parameters, worst rel L2: " + ", ".join(f"{r[0]} {r[3]:.2e}" for r in worst))
    assert not bad, [(r[0], r[1], f"rel_l2 {r[3]:.3e}", f"frac {r[4]:.4f}") for r in bad[:20]]
    assert sum(1 for r in report if GC.small_tensor(r[0])) > 100  # the small tensors ARE in the table
    # AdamW step 1 moves every parameter by lr * g / (|g| + eps): parameters whose gradient is ~eps (1e-8) amplify
    # round-off differences, up to 2 * lr for a sign flip; everything else must agree
    d = (got_p - ref_p).abs()
    pf = param_frac if param_frac is not None else (1e-3 if rel_l2 == GC.REL_L2 else 2e-2)
    assert d.max() <= 2.1e-4 and float((d > 1e-5).float().mean()) < pf, (float(d.max()), float((d > 1e-5).float().mean()))


FLIP_REL_L2, FLIP_FRAC = 5e-2, 1.0  # un-frozen R50 step: ONE discrete event - an attention-mask cell or an FFN unit within round-off
# of its threshold - moves the gradients of the decoder layer around it by 0.3 - 1 % (seen: layer 7 at 3.7e-3 / 45 % of the entries,
# from a 1e-6 wobble of res5 between two EAGER steps).  Since round 4 the forward pass is bit-reproducible from run to run
# (tools/probe_forward_bits.py: the library's atomics-based stride-2 / VGGish kernels and the SEM pool's float atomics are gone),
# so no such event separates an eager from a replayed step any more; the bound stays as a guard for library versions that
# reintroduce one.  The strict bound is applied with the discrete choices frozen.


def test_graphed_step_equals_eager_step(rig):
    model, opt, batches, state = rig
    graphed, out = GC.eager_and_graphed(model, opt, batches)
    for i, (ref_l, got_l, report, ref_p, got_p, _, _) in enumerate(out):
        _check_pair(i, ref_l, got_l, report, ref_p, got_p, rel_l2=FLIP_REL_L2, frac=FLIP_FRAC)
    assert len(graphed.graphs) == 1  # both batches share one signature -> one capture, second batch via static buffers
    assert any(abs(out[1][1][k] - out[0][0][k]) > 1e-4 for k in out[1][1])  # the second batch really went through


def test_graphed_step_equals_eager_step_frozen_choices(rig):
    """the strict per-parameter bound (5e-3 relative L2, 25 % of the entries; the backbones' library gradients 2e-2 / 80 %) with
    the discrete choices of a recorded step injected into both runs: no mask cell, matching or sampled point can differ"""
    model, opt, batches, state = rig
    GC.freeze_choices(model, opt, batches[0])
    try:
        graphed, out = GC.eager_and_graphed(model, opt, batches[:1])
    finally:
        GC.unfreeze_choices(model)
    ref_l, got_l, report, ref_p, got_p, _, _ = out[0]
    _check_pair(0, ref_l, got_l, report, ref_p, got_p)
    assert len(graphed.graphs) == 1


def test_graphed_step_equals_eager_step_pvt_recipe():
    """the same per-parameter comparison on the PVTv2-B5 recipe (bf16 backbones: ~630 bias gradients through the deferred
    grouped column sums, the pre-norm kernels, stochastic depth off).  The discrete choices of a recorded step are injected into
    both runs (GC.freeze_choices): un-frozen, bf16 round-off in the backbones flips attention-mask cells between an eager
    and a replayed step and whole gradient rows move by 3-8 % (measured: median relative L2 2.8e-2 over the 2 438 parameters)."""
    with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):  # (MIOpen: see the note above PVT_LIB_FRAC)
        model, opt, batches, state = GC.build("pvt")
        GC.freeze_choices(model, opt, batches[0])
        try:
            graphed, out = GC.eager_and_graphed(model, opt, batches[:1])
        finally:
            GC.unfreeze_choices(model)
    ref_l, got_l, report, ref_p, got_p, _, _ = out[0]
    # head: the R50 recipe's strict bound; the bf16 backbones' parameters: relative L2 <= 2e-2 (measured <= 5.3e-3), and the AdamW
    # sign-flip share of the update among their ~160 M entries <= 2 %
    _check_pair(0, ref_l, got_l, report, ref_p, got_p, lib_frac=PVT_LIB_FRAC, param_frac=2e-2)
    assert len(graphed.graphs) == 1


def test_per_parameter_comparison_catches_a_bypassed_colsum():
    """The comparison above must FAIL on the build of rounds 1-2: ops.colsum replaced by ATen's split reductions (memset nodes
    inside the captured step) with the runtime's graph packet capture on.  Run in a child process (the runtime reads the
    switch at its first HIP call); skipped when this runtime replays memset nodes correctly (then there is nothing to catch)."""
    import json
    import subprocess
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE="1", COMBO_ALLOW_PACKET_CAPTURE="1")
    # --single-stream: since round 6 the two encoders run on two streams and the captured step has parallel branches - the runtime
    # replays the memset nodes of THAT graph correctly (measured: no parameter flagged); the hazard is the linear graph's
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "graph_compare.py"), "r50", "--bypass-colsum", "--single-stream"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    if res["selftest"]:
        pytest.skip("this runtime replays hipGraph memset nodes correctly")
    assert res["captured"]
    assert res["failed_small"], res  # biases / level embeddings named one by one
    print(f"[bypassed colsum] {len(res['failed'])} of {res['n']} parameters flagged, e.g. {res['failed_small'][:4]}")


def test_fusion_dropout_is_fresh_on_every_replay():
    from combo_avs_amd.ops import bifuse
    torch.manual_seed(0)
    B, N, C, H = 2, 784, 256, 8
    dev = "cuda"
    x = torch.randn(B, N, C, device=dev)
    args = (torch.ones(C, device=dev), torch.zeros(C, device=dev), 1e-5, torch.randn(N, C, device=dev) * 0.1,
            torch.randn(B, H, C, device=dev) * 0.05, torch.zeros(B, H, device=dev), torch.randn(B, H, C, device=dev),
            torch.zeros(C, device=dev), torch.full((C,), 0.5, device=dev))
    counter = bifuse.step_counter(torch.device("cuda", torch.cuda.current_device()))
    bifuse.token_op(x, *args, 0.1, seed=77)  # lazy init outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        counter.add_(1)
        y, pooled, spa = bifuse.token_op(x, *args, 0.1, seed=77)
    outs = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        outs.append(y.clone())
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    # same counter value + same seed -> same masks (the backward regenerates them from exactly this pair)
    c = counter.clone()
    y1, _, _ = bifuse.token_op(x, *args, 0.1, seed=77)
    counter.copy_(c)
    y2, _, _ = bifuse.token_op(x, *args, 0.1, seed=77)
    assert torch.equal(y1, y2)


def test_graphed_training_actually_learns(rig):
    """40 replays of the captured step on one batch: the optimiser updates the flat parameter buffer in place and the graph
    must see the new values (a capture that froze copies of the weights would leave the loss where it started)."""
    from combo_avs_amd.trainer import GraphedTrainStep
    model, opt, batches, state = rig
    _reset = GC.reset
    snap = opt.flat_param.clone()
    _reset(opt, snap)
    old = [(s[0], s[1], s[2], s[3]) for s in opt.segments]
    for s in opt.segments:
        s[2] = s[2] * 20.0  # lr x 20 for the test (clip 0.01 keeps every step tiny at the reference's 1e-4 / 1e-5)
    try:
        step = GraphedTrainStep(model, opt)
        totals = []
        for _ in range(40):
            losses = step(batches[0])
            totals.append(float(sum(losses.values())))
        assert all(t == t for t in totals)
        # (at 20 x the learning rate the curve is noisy and, through the atomics of the library kernels, differs from run to run:
        # 85.1 -> 75.0 +- 0.2 after 10 steps in every run, 71.8 ... 78.8 after 40 (tools/learn_probe.py); a capture that froze
        # the weights would stay at 85.1)
        assert min(totals[5:]) < 0.95 * totals[0], (totals[0], min(totals[5:]), totals[-1])
        assert len(step.graphs) == 1
    finally:
        for s, o in zip(opt.segments, old):
            s[2] = o[2]
        _reset(opt, snap)


def test_backbone_weight_gradients_land_in_the_flat_buffer_without_a_concatenation(rig):
    """backbone._FoldAll registers its FrozenBN-folded weights as aliases of their parameters (ops.linear.register_grad_aliases):
    the weight-gradient kernels and the fold's backward write the optimiser's flat gradient buffer directly.  (a) the flat
    buffer after a backward pass equals the one the concatenation path (GRAD_IN_PLACE = False) fills, per parameter, within the
    run-to-run noise of two backward passes (graph_compare's bounds); (b) `aten.cat` moves > 150 MB
    less than on that path (the two ResNet-50s' 188 MB of weight gradients; what still goes through it: audio_mlp's and a few
    other library-computed gradients of the head)."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from combo_avs_amd import backbone as B
    model, opt, batches, state = rig
    snap = opt.flat_param.clone()

    class CatBytes(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.n = 0

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            if func.__name__.split(".")[0] == "cat" and torch.is_tensor(out):
                self.n += out.numel() * out.element_size()
            return out

    def grads(in_place):
        GC.reset(opt, snap)
        old, B.GRAD_IN_PLACE = B.GRAD_IN_PLACE, in_place
        try:
            losses = model(batches[0])
            total = sum(losses.values())
            opt.flat_grad.zero_()
            with CatBytes() as spy:
                opt.backward(total)
            torch.cuda.synchronize()
            return opt.flat_grad.clone(), spy.n
        finally:
            B.GRAD_IN_PLACE = old

    g_cat, bytes_cat = grads(False)
    g_inp, bytes_inp = grads(True)
    assert bytes_cat - bytes_inp > 150e6, (bytes_inp, bytes_cat)  # 2 x 23.5 M ResNet-50 weights no longer pass through torch.cat
    report = GC.per_parameter(opt, g_inp, g_cat)
    assert not GC.failures(report), GC.failures(report)[:10]  # (two backward passes differ by ~1e-5: atomics in the library kernels)
    # (c) against plain autograd - no flat-buffer targets, no deferred grouped launch, every gradient a fresh tensor: the whole
    # in-place machinery (dense layers since round 2, 1x1 convolutions and folded backbone weights since round 5) changes WHERE a
    # gradient is written, not its value (summation orders of the grouped launch differ: round-off only)
    GC.reset(opt, snap)
    total = sum(model(batches[0]).values())
    plain = torch.autograd.grad(total, opt.params, allow_unused=True)
    g_plain = torch.zeros_like(opt.flat_grad)
    for g, off, p in zip(plain, opt.offsets, opt.params):
        if g is not None:
            g_plain[off:off + p.numel()].copy_(g.reshape(-1))
    report = GC.per_parameter(opt, g_inp, g_plain)
    assert not GC.failures(report), GC.failures(report)[:10]
    GC.reset(opt, snap)
