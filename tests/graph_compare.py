"""Eager vs hipGraph-replayed training step, compared PER NAMED PARAMETER (test helper + command line).

Round 3's whole-buffer comparison (1 % of 87 M gradient entries may be off) could not see the tensors that the hipGraph
memset-node bug corrupted for two rounds: every bias and level embedding together is far below 1 % of the flat buffer.  Here
every parameter of the optimiser's table gets its own verdict:

    rel_l2  = ||g_graph - g_eager|| / ||g_eager||                       (0 / 0 = 0; x / 0 = inf)
    frac    = share of entries beyond 2e-3 RMS(g_eager) + 2e-3 |g_eager|

    python tests/graph_compare.py r50|pvt [--bypass-colsum]    -> one JSON line {"failed": [...], "worst": [...], "n": N}

--bypass-colsum replaces ops.colsum's kernels by ATen's `sum` (split reductions = memset nodes inside the captured step): with
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 that is the build of rounds 1-2, and the comparison has to fail on it
(tests/test_graph_gpu.py::test_per_parameter_comparison_catches_a_bypassed_colsum)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REL_L2, FRAC = 5e-3, 0.25  # the head's parameters (own, deterministic kernels): measured eager-vs-replay noise <= 1e-4 / 0 %
# the backbones' parameters: the stride-2 / 7x7 weight gradients are MIOpen kernels whose summation order changes from run to run
# (measured on *.stem.conv1.weight: up to 5.2e-3 / 51 % between an eager and a replayed step) and every other backbone gradient
# inherits a little of it through the input gradients
LIB_REL_L2, LIB_FRAC = 2e-2, 0.8


def _to_gpu(batch):
    return [{k: (v.cuda() if torch.is_tensor(v) else [{kk: vv.cuda() for kk, vv in i.items()} for i in v])
             for k, v in b.items()} for b in batch]


def build(recipe):
    """model (no dropout / stochastic depth: eager and replayed steps must agree number for number), optimiser, two batches"""
    sys.path.insert(0, ROOT)
    import combo_avs_amd  # noqa: F401
    from bench import synth_batch
    from combo_avs_amd import combo_cfg
    from combo_avs_amd.meta_arch import build_model
    from combo_avs_amd.trainer import FlatAdamW
    yaml = {"r50": "configs/avs_s4/COMBO_R50_bs8_90k.yaml", "pvt": "configs/avs_s4/COMBO_PVTV2B5_bs8_90k.yaml"}[recipe]
    cfg = combo_cfg(os.path.join(ROOT, yaml))
    torch.manual_seed(0)
    model = build_model(cfg).cuda().train()
    if recipe == "pvt":
        model.backbone_dtype = torch.bfloat16
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if type(m).__name__ == "DropPath":
            m.p = 0.0
    for a in model.sem_seg_head.fusion_module.b_attn.attn_list:
        a.dropout = 0.0
    bank = torch.rand(40_000_000, generator=torch.Generator().manual_seed(5)).cuda()
    state = {"off": 0}

    def point_source(n, p):
        o = state["off"]
        state["off"] = o + n * p * 2
        return bank[o:o + n * p * 2].view(n, p, 2)

    model.criterion.point_source = point_source
    # every forward (eager, the capture's warm-up iterations, the capture itself) reads the bank from offset 0
    model.register_forward_pre_hook(lambda m, a: state.__setitem__("off", 0))
    opt = FlatAdamW(model, base_lr=1e-4, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01)
    batches = [_to_gpu(synth_batch(2, 5, 224, 224, "cpu", seed=s)) for s in (11, 12)]
    return model, opt, batches, state


def reset(opt, snap):
    opt.flat_param.copy_(snap)
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    opt.step_count = 0


def per_parameter(opt, g, ref_g):
    """-> [(name, numel, rms_ref, rel_l2, frac_beyond)] for every entry of the optimiser's parameter table"""
    rows = []
    stats = []
    for (p, name, _, _), off in zip(opt.entries, opt.offsets):
        n = p.numel()
        a, b = g[off:off + n].double(), ref_g[off:off + n].double()
        rms = b.pow(2).mean().sqrt()
        err = (a - b).abs()
        stats.append(torch.stack([rms, (a - b).norm(), b.norm(), (err > 2e-3 * rms + 2e-3 * b.abs()).double().mean()]))
        rows.append((name, n))
    st = torch.stack(stats).cpu()  # one copy for the whole table
    out = []
    for (name, n), (rms, dn, bn, frac) in zip(rows, st.tolist()):
        rel = 0.0 if dn == 0.0 else (dn / bn if bn > 0 else float("inf"))
        out.append((name, n, rms, rel, frac))
    return out


def failures(report, rel_l2=REL_L2, frac=FRAC, lib_frac=None):
    """parameters beyond the bound.  A gradient that is mathematically zero (the bilateral fusion's v_proj bias cancels in its
    soft-max: RMS 1e-13 against 1e-2 .. 1e-1 elsewhere) has no relative error: such a tensor only has to stay below 1e-6 of
    the table's typical RMS."""
    typical = sorted(r[2] for r in report)[len(report) // 2]
    floor = 1e-6 * typical
    out = []
    for r in report:
        if r[2] < floor:
            if r[3] != float("inf") and r[3] * r[2] < floor * 10:  # ||got - ref|| / sqrt(n) stays tiny as well
                continue
        lib = r[0].startswith(("backbone.", "pre_sam_backbone."))
        # lib_frac: the entry fraction allowed for the backbones' parameters (bf16 recipe: a 2e-3 element-wise bound means nothing
        # against bf16 round-off - 83 % of the entries of a 64-entry norm weight - so that recipe passes 1.0 and keeps the L2 bound)
        lf = max(frac, LIB_FRAC) if lib_frac is None else lib_frac
        if not (r[3] <= (max(rel_l2, LIB_REL_L2) if lib else rel_l2) and r[4] <= (lf if lib else frac)):
            out.append(r)
    return out


def small_tensor(name):
    """the tensors the memset-node bug hit: biases, level / position embeddings, norm affines"""
    return name.endswith(".bias") or "level_embed" in name or "norm" in name or "query_embed" in name or "audio_pos" in name


def freeze_choices(model, opt, batch):
    """One recorded eager step on `batch`; its discrete choices - the 9 attention masks, the Hungarian pairs of all 10 outputs, the
    importance-sampled point coordinates - are then injected into every later step (model hooks attn_mask_override /
    criterion.frozen_choices): eager and replayed steps become the SAME continuous function of the weights, so bf16 round-off
    in the backbones (its summation order differs between an eager and a captured library GEMM) can no longer flip a mask
    cell or a matching and move whole gradient rows by per cents."""
    from combo_avs_amd.trainer import train_step
    dec, crit = model.sem_seg_head.predictor, model.criterion
    snap = opt.flat_param.clone()
    dec.record_attn_masks, crit.record_choices = [], True
    try:
        train_step(model, opt, batch)
        torch.cuda.synchronize()
        masks = list(dec.record_attn_masks)
        src_q, tgt_g, _ = crit.last_indices
        frozen = {"match_src": src_q.clone(), "match_tgt": tgt_g.clone(), "coords": crit.last_coords.clone()}
    finally:
        dec.record_attn_masks, crit.record_choices = None, False
    reset(opt, snap)
    assert len(masks) == dec.num_layers
    dec.attn_mask_override, crit.frozen_choices = masks, frozen


def unfreeze_choices(model):
    model.sem_seg_head.predictor.attn_mask_override = None
    model.criterion.frozen_choices = None


def eager_and_graphed(model, opt, batches):
    """-> per batch: (eager losses, graphed losses, per-parameter report, eager / graphed parameters after the update)"""
    from combo_avs_amd.trainer import GraphedTrainStep, train_step
    snap = opt.flat_param.clone()
    eager = []
    for b in batches:
        reset(opt, snap)
        losses = train_step(model, opt, b)
        eager.append(({k: float(v) for k, v in losses.items()}, opt.flat_grad.clone(), opt.flat_param.clone()))
    graphed = GraphedTrainStep(model, opt)
    out = []
    for i, b in enumerate(batches):
        reset(opt, snap)
        losses = graphed(b)
        ref_l, ref_g, ref_p = eager[i]
        out.append((ref_l, {k: float(v) for k, v in losses.items()}, per_parameter(opt, opt.flat_grad, ref_g),
                    ref_p, opt.flat_param.clone(), ref_g, opt.flat_grad.clone()))
    reset(opt, snap)
    return graphed, out


def main():
    recipe = sys.argv[1]
    if "--bypass-colsum" in sys.argv:
        import combo_avs_amd  # noqa: F401
        from combo_avs_amd.ops import colsum

        def aten_channel_sum(x, A, C, L, out_dtype=None):
            return x.view(A, C, L).sum((0, 2), dtype=torch.float32).to(out_dtype or x.dtype)
        colsum.channel_sum = aten_channel_sum
        colsum.DEFER = False
    if "--single-stream" in sys.argv:  # the launch order of rounds 1-5: one stream, a linear graph (the form whose memset nodes the
        import combo_avs_amd  # noqa: F401  # runtime's packet capture replays wrongly; a graph with parallel branches takes another path)
        from combo_avs_amd.meta_arch import MaskFormer
        MaskFormer.parallel_backbones = MaskFormer.parallel_audio = False
    model, opt, batches, _ = build(recipe)
    from combo_avs_amd.trainer import graph_memset_selftest
    selftest = graph_memset_selftest(torch.device("cuda", 0))
    if "--frozen" in sys.argv:
        batches = batches[:1]
        freeze_choices(model, opt, batches[0])
    graphed, out = eager_and_graphed(model, opt, batches)
    bad = []
    for _, _, rep, *_ in out:
        bad += failures(rep)
    worst = sorted(out[0][2], key=lambda r: -r[3])[:8]
    if "--dump" in sys.argv:
        with open(sys.argv[sys.argv.index("--dump") + 1], "w") as f:
            json.dump([o[2] for o in out], f)
    print(json.dumps({"selftest": selftest, "captured": bool(graphed.graphs), "n": len(out[0][2]),
                      "failed": sorted({r[0] for r in bad}), "failed_small": sorted({r[0] for r in bad if small_tensor(r[0])}),
                      "worst": [(r[0], r[1], r[3], r[4]) for r in worst]}))


if __name__ == "__main__":
    main()
