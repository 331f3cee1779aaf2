#!/usr/bin/env python3
"""Which part of the training step survives hipGraph capture?  `python tools/graph_bisect.py <stage>` captures one
stage after two eager runs and replays it twice; run every stage in its own process (a failing capture can segfault).
Stages: vggish bb_fwd bb_bwd semmix pd_fwd pd_bwd fuse_bwd dec_fwd dec_bwd crit_bwd full"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW
from combo_avs_amd.modeling.semmix import sem_mix
from bench import synth_batch

stage = sys.argv[1]
clips = int(os.environ.get("CLIPS", "2"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_R50_bs8_90k.yaml"))
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
if os.environ.get("DTYPE", "bf16") == "bf16":
    model.backbone_dtype = torch.bfloat16
opt = FlatAdamW(model)
batch = synth_batch(clips, 5, 224, 224, dev, 1)
head = model.sem_seg_head
amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=os.environ.get("DTYPE", "bf16") == "bf16")
images = torch.cat([b["images"] for b in batch]).float()
images = (images - model.pixel_mean) / model.pixel_std
pre = torch.cat([b["pre_masks"] for b in batch]).float()
pre = (pre - model.pixel_mean) / model.pixel_std
mel = torch.cat([b["audio_log_mel"] for b in batch])


def P(mod):
    return [p for p in mod.parameters() if p.requires_grad]


def grads(outs, params):
    outs = [o for o in outs if o.requires_grad]
    return torch.autograd.grad([o.float().sum() for o in outs], params, allow_unused=True)


with torch.no_grad(), amp:
    audio0 = model.audio_backbone(mel).float().unsqueeze(1)
    f0 = model.backbone(images)
    p0 = model.pre_sam_backbone(pre)
    feats0 = {k: v.float() for k, v in sem_mix(f0, p0, model.scale_factor_module).items()}
with torch.no_grad():
    mf0, _, ms0 = head.pixel_decoder.forward_features(feats0)
    fused0 = head.fusion_module({"res2": mf0}, audio0)
    a2560 = head.audio_transformation(fused0["audio"])
    out0 = head.predictor(ms0, a2560, fused0["visual"]["res2"], None)
targets = model.prepare_targets([i for b in batch for i in b["instances"]], images)
model.criterion.num_masks_override = torch.tensor([float(sum(len(t["labels"]) for t in targets))], device=dev)


def req(x):
    return x.detach().clone().requires_grad_(True)


def run():
    if stage == "vggish":
        with torch.no_grad(), amp:
            return model.audio_backbone(mel).float()
    if stage == "bb_fwd":
        with torch.no_grad(), amp:
            return model.backbone(images)["res5"]
    if stage == "bb_bwd":
        with amp:
            f = model.backbone(images)
        return grads(list(f.values()), P(model.backbone))[0]
    if stage == "semmix":
        f = {k: req(v) for k, v in f0.items()}
        o = sem_mix(f, p0, model.scale_factor_module)
        return grads(list(o.values()), list(f.values()) + P(model.scale_factor_module))[0]
    if stage == "pd_fwd":
        with torch.no_grad():
            return head.pixel_decoder.forward_features(feats0)[0]
    if stage == "pd_bwd":
        mf, _, ms = head.pixel_decoder.forward_features(feats0)
        return grads([mf] + list(ms), P(head.pixel_decoder))[0]
    if stage == "fuse_bwd":
        fused = head.fusion_module({"res2": req(mf0)}, audio0)
        a256 = head.audio_transformation(fused["audio"])
        return grads([fused["visual"]["res2"], a256], P(head.fusion_module) + P(head.audio_transformation))[0]
    if stage == "dec_fwd":
        with torch.no_grad():
            return head.predictor(ms0, a2560, fused0["visual"]["res2"], None)["pred_masks"]
    if stage == "dec_bwd":
        out = head.predictor(ms0, a2560, fused0["visual"]["res2"], None)
        leaves = [out["pred_logits"], out["pred_masks"]] + [t for a in out["aux_outputs"] for t in a.values()] + out["middles_attn_mask"]
        return grads(leaves, P(head.predictor))[0]
    if stage == "crit_bwd":
        o = {"pred_logits": req(out0["pred_logits"]), "pred_masks": req(out0["pred_masks"]),
             "aux_outputs": [{k: req(v) for k, v in a.items()} for a in out0["aux_outputs"]],
             "middles_attn_mask": [req(m) for m in out0["middles_attn_mask"]]}
        losses = model.criterion(o, targets)
        total = torch.stack(list(losses.values())).sum()
        return torch.autograd.grad(total, [o["pred_masks"]])[0]
    if stage == "full":
        losses = model(batch)
        total = torch.stack(list(losses.values())).sum()
        opt.backward(total)
        return total
    raise SystemExit("unknown stage " + stage)


if os.environ.get("PROFILE"):
    # kernel inventory of one stage (eager): launches, GPU time, top kernels
    from torch.profiler import profile, ProfilerActivity
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    if os.environ.get("STACKS"):  # which call sites own the copies / adds
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            run()
            torch.cuda.synchronize()
        rows = []
        for e in prof.key_averages(group_by_stack_n=8):
            if e.key in os.environ["STACKS"].split(",") and e.device_time_total > 0:
                rows.append(e)
        for e in sorted(rows, key=lambda e: -e.device_time_total)[:int(os.environ.get("TOP", "25"))]:
            st = [f for f in e.stack if "combo" in f or "bench" in f or "tools/" in f][:3]
            print(f"{e.device_time_total / 1e3:7.3f} ms {e.count:4d}x {e.key:18s} | " + " <- ".join(x.split("/")[-1][:60] for x in st))
        raise SystemExit(0)
    if os.environ.get("OPS"):  # aten-op level table (which framework ops own the small kernels)
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            run()
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=bool(os.environ.get("SHAPES"))).table(
            sort_by="self_cuda_time_total", row_limit=int(os.environ.get("TOP", "40")), max_name_column_width=48,
            max_shapes_column_width=60))
        raise SystemExit(0)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.device_type.name != "CPU" or e.self_device_time_total > 0]
    n = sum(e.count for e in ev)
    t = sum(e.self_device_time_total for e in ev)
    print(f"[{stage}] PROFILE launches={n} gpu_ms={t / 1e3:.2f}")
    for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:int(os.environ.get("TOP", "14"))]:
        print(f"    {e.self_device_time_total / 1e3:7.3f} ms {e.count:5d}x  {e.key[:110]}")
    raise SystemExit(0)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        run()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print(f"[{stage}] eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
print(f"[{stage}] captured", flush=True)
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
print(f"[{stage}] REPLAY OK finite={bool(torch.isfinite(out.float()).all())}", flush=True)
