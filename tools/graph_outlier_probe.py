#!/usr/bin/env python3
"""Round 6: the eager-vs-replay outlier of the PVT recipe (tests/test_graph_gpu.py: one run in twelve, every parameter of the last
decoder layer off by 1.1e-1).  One process, frozen discrete choices, every run from the same parameters:

    E0..E3   four EAGER steps                               -> eager-vs-eager spread
    for c in captures: a FRESH GraphedTrainStep, `replays` replays each
    one more eager step at the end

    python tools/graph_outlier_probe.py pvt|r50 [captures=8] [replays=2] [out.json]

All gradient buffers are kept; the output is (1) per run the relative L2 against E0 per module group, (2) the pairwise distance
matrix of the runs over the HEAD's parameters (runs that land in the same "state" are 0 apart when the kernels between them are
deterministic), (3) for the two most distant runs the parameters that differ most.
COMBO_PROBE_SWITCH=sra,inplace,classheads,prenorm,defer,fp32bb,ffnfuse turns sra.ENABLED / backbone.GRAD_IN_PLACE / BATCH_CLASS_HEADS /
backbone_pvt.PRENORM / colsum.DEFER off, runs the backbones in fp32, un-fuses the FFN's ReLU gradient."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import graph_compare as GC  # noqa: E402


def group_of(name):
    m = re.match(r"sem_seg_head\.predictor\.transformer_(self_attention|cross_attention|ffn)_layers\.(\d+)\.", name)
    if m:
        return f"dec{m.group(2)}"
    if name.startswith("sem_seg_head.predictor."):
        return "dec_other"
    if name.startswith("sem_seg_head.pixel_decoder."):
        return "pixdec"
    if name.startswith("sem_seg_head."):
        return "head_other"
    return name.split(".")[0]


def main():
    recipe = sys.argv[1] if len(sys.argv) > 1 else "pvt"
    captures = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    replays = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    sw = [s for s in os.environ.get("COMBO_PROBE_SWITCH", "").split(",") if s]
    import combo_avs_amd  # noqa: F401
    if "sra" in sw:
        from combo_avs_amd.ops import sra
        sra.ENABLED = False
    if "inplace" in sw:
        from combo_avs_amd import backbone
        backbone.GRAD_IN_PLACE = False
    if "classheads" in sw:
        from combo_avs_amd.modeling import transformer_decoder as TD
        TD.BATCH_CLASS_HEADS = False
    if "prenorm" in sw:
        from combo_avs_amd import backbone_pvt
        backbone_pvt.PRENORM = False
    if "defer" in sw:
        from combo_avs_amd.ops import colsum
        colsum.DEFER = False
    if "ffnfuse" in sw:
        from combo_avs_amd.ops import linear
        linear.FFN_FUSED_RELU_GRAD = False
    if "det" in sw:  # MIOpen: deterministic solvers only (tools/pvt_forward_states.py: the default immediate-mode solver of the
        torch.backends.cudnn.deterministic = True  # PVT spatial-reduction convolutions is not run-to-run reproducible)
    if "find" in sw:
        torch.backends.cudnn.benchmark = True
    from combo_avs_amd.trainer import GraphedTrainStep, train_step
    model, opt, batches, state = GC.build(recipe)
    if "fp32bb" in sw:
        model.backbone_dtype = torch.float32
    b = batches[0]
    GC.freeze_choices(model, opt, b)
    snap = opt.flat_param.clone()
    runs = []  # (tag, grad, losses)

    def eager(tag):
        GC.reset(opt, snap)
        losses = train_step(model, opt, b)
        torch.cuda.synchronize()
        runs.append((tag, opt.flat_grad.clone(), {k: float(v) for k, v in losses.items()}))

    for i in range(4):
        eager(f"E{i}")
    for c in range(captures):
        graphed = GraphedTrainStep(model, opt)
        for r in range(replays):
            GC.reset(opt, snap)
            losses = graphed(b)
            torch.cuda.synchronize()
            runs.append((f"G{c}.{r}", opt.flat_grad.clone(), {k: float(v) for k, v in losses.items()}))
        assert len(graphed.graphs) == 1
        del graphed
        torch.cuda.synchronize()
    eager("Elast")

    names = [n for (_, n, _, _) in opt.entries]
    sizes = [p.numel() for (p, _, _, _) in opt.entries]
    offs = list(opt.offsets)
    gkeys = sorted({group_of(n) for n in names})
    ref = runs[0][1]
    # a gradient that is mathematically zero (the fusion's v_proj bias cancels in its soft-max) has no relative error
    rms = torch.stack([ref[o:o + n].double().pow(2).mean().sqrt() for o, n in zip(offs, sizes)]).cpu()
    typical = rms.sort().values[len(rms) // 2]
    live = [bool(r > 1e-6 * typical) for r in rms]
    log = {"switches": sw, "recipe": recipe, "runs": []}
    for tag, g, losses in runs[1:]:
        num = {k: 0.0 for k in gkeys}
        den = {k: 0.0 for k in gkeys}
        worst = []
        for n, o, s, lv in zip(names, offs, sizes, live):
            if not lv:
                continue
            a, bb = g[o:o + s].double(), ref[o:o + s].double()
            d2, b2 = float((a - bb).pow(2).sum()), float(bb.pow(2).sum())
            k = group_of(n)
            num[k] += d2
            den[k] += b2
            worst.append(((d2 / max(b2, 1e-300)) ** 0.5, n))
        worst.sort(reverse=True)
        gr = {k: (num[k] / max(den[k], 1e-300)) ** 0.5 for k in gkeys}
        dl = max(abs(losses[k] - runs[0][2][k]) / (abs(runs[0][2][k]) + 1e-9) for k in losses)
        print(f"[{tag} vs E0] loss {dl:.1e} | " + " ".join(f"{k}={v:.1e}" for k, v in gr.items()), flush=True)
        print(f"[{tag} vs E0]    worst: " + ", ".join(f"{n.replace('sem_seg_head.', '')} {v:.1e}" for v, n in worst[:4]), flush=True)
        log["runs"].append({"tag": tag, "groups": gr, "loss_diff": dl, "worst": worst[:10]})

    # pairwise distances over the head's parameters
    head = [(o, s) for n, o, s, lv in zip(names, offs, sizes, live) if lv and n.startswith("sem_seg_head.")]
    hv = [torch.cat([g[o:o + s] for o, s in head]).double() for _, g, _ in runs]
    nrm = hv[0].norm()
    D = [[float((x - y).norm() / nrm) for y in hv] for x in hv]
    tags = [t for t, _, _ in runs]
    print("pairwise relative L2 over the head's parameters (x 1e-3):")
    print("        " + " ".join(f"{t:>6}" for t in tags))
    for t, row in zip(tags, D):
        print(f"{t:>6}  " + " ".join(f"{1e3 * v:6.2f}" for v in row))
    log["tags"], log["head_distance"] = tags, D
    # the two most distant runs: which parameters
    bi, bj = max(((i, j) for i in range(len(runs)) for j in range(i)), key=lambda ij: D[ij[0]][ij[1]])
    gi, gj = runs[bi][1], runs[bj][1]
    rows = []
    for n, o, s, lv in zip(names, offs, sizes, live):
        if lv:
            a, bb = gi[o:o + s].double(), gj[o:o + s].double()
            rows.append((float((a - bb).norm() / bb.norm().clamp_min(1e-300)), n, s))
    rows.sort(reverse=True)
    print(f"most distant pair: {tags[bi]} vs {tags[bj]} ({D[bi][bj]:.2e}); parameters:")
    for v, n, s in rows[:25]:
        print(f"   {v:.2e}  {n} [{s}]")
    log["most_distant"] = {"pair": (tags[bi], tags[bj]), "params": rows[:60]}
    if out_path:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(log, f)


if __name__ == "__main__":
    main()
