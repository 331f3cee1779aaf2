#!/usr/bin/env python3
"""Wall-time (GPU-synchronised) of the sections of one training step at BASELINE config 2 (bs=8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW
from bench import synth_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_R50_bs8_90k.yaml"))
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
model.backbone_dtype = torch.bfloat16
opt = FlatAdamW(model)
batch = synth_batch(8, 5, 224, 224, dev, 1)


class T:
    def __init__(self): self.t = {}
    def section(self, name):
        outer = self
        class C:
            def __enter__(s): torch.cuda.synchronize(); s.t0 = time.perf_counter()
            def __exit__(s, *a): torch.cuda.synchronize(); outer.t[name] = outer.t.get(name, 0) + time.perf_counter() - s.t0
        return C()

head = model.sem_seg_head
for it in range(8):
    tm = T()
    with tm.section("total"):
        with tm.section("inputs+vggish"):
            images = torch.cat([b["images"] for b in batch]).float()
            images = (images - model.pixel_mean) / model.pixel_std
            pre = torch.cat([b["pre_masks"] for b in batch]).float()
            pre = (pre - model.pixel_mean) / model.pixel_std
            mel = torch.cat([b["audio_log_mel"] for b in batch])
            amp = torch.autocast("cuda", dtype=torch.bfloat16)
            with torch.no_grad(), amp:
                audio = model.audio_backbone(mel).float().unsqueeze(1)
        with tm.section("backbones fwd"):
            with amp:
                f = model.backbone(images)
                p = model.pre_sam_backbone(pre)
        with tm.section("sem mix"):
            from combo_avs_amd.modeling.semmix import sem_mix
            feats = sem_mix(f, p, model.scale_factor_module)
        with tm.section("pixel decoder fwd"):
            mf, _, ms = head.pixel_decoder.forward_features(feats)
        with tm.section("avfuse+audio mlp fwd"):
            fused = head.fusion_module({"res2": mf}, audio)
            a256 = head.audio_transformation(fused["audio"])
        with tm.section("decoder fwd"):
            out = head.predictor(ms, a256, fused["visual"]["res2"], None)
        with tm.section("criterion fwd"):
            targets = model.prepare_targets([i for b in batch for i in b["instances"]], images)
            losses = model.criterion(out, targets)
            total = torch.stack([v * model.criterion.weight_dict[k] for k, v in losses.items()]).sum()
        with tm.section("backward"):
            opt.backward(total)
        with tm.section("optimizer"):
            opt.all_reduce_grads(); opt.step()
    if it >= 3:
        print(" | ".join(f"{k} {v * 1e3:.1f}" for k, v in tm.t.items()))

# ---- torch profiler over the backward pass only ----
if os.environ.get("PROFILE_BWD"):
    from torch.profiler import profile, ProfilerActivity
    images = torch.cat([b["images"] for b in batch]).float()
    losses = model(batch)
    total = torch.stack(list(losses.values())).sum()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        opt.backward(total)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))

if os.environ.get("BWD_STAGES"):
    def P(mod): return [p for p in mod.parameters() if p.requires_grad]
    for it in range(4):
        images = torch.cat([b["images"] for b in batch]).float(); images = (images - model.pixel_mean) / model.pixel_std
        pre = torch.cat([b["pre_masks"] for b in batch]).float(); pre = (pre - model.pixel_mean) / model.pixel_std
        mel = torch.cat([b["audio_log_mel"] for b in batch])
        amp = torch.autocast("cuda", dtype=torch.bfloat16)
        with torch.no_grad(), amp:
            audio = model.audio_backbone(mel).float().unsqueeze(1)
        with amp:
            f = model.backbone(images); p = model.pre_sam_backbone(pre)
        feats = sem_mix(f, p, model.scale_factor_module)
        feats_d = {k: v.detach().requires_grad_(True) for k, v in feats.items()}
        mf, _, ms = head.pixel_decoder.forward_features(feats_d)
        mf_d = mf.detach().requires_grad_(True); ms_d = [m.detach().requires_grad_(True) for m in ms]
        fused = head.fusion_module({"res2": mf_d}, audio); a256 = head.audio_transformation(fused["audio"])
        fv_d = fused["visual"]["res2"].detach().requires_grad_(True); a256_d = a256.detach().requires_grad_(True)
        out = head.predictor(ms_d, a256_d, fv_d, None)
        leaves = [out["pred_logits"], out["pred_masks"]] + [t for a in out["aux_outputs"] for t in a.values()] + out["middles_attn_mask"]
        out_d = {"pred_logits": out["pred_logits"].detach().requires_grad_(True), "pred_masks": out["pred_masks"].detach().requires_grad_(True),
                 "aux_outputs": [{k: v.detach().requires_grad_(True) for k, v in a.items()} for a in out["aux_outputs"]],
                 "middles_attn_mask": [m.detach().requires_grad_(True) for m in out["middles_attn_mask"]]}
        leaves_d = [out_d["pred_logits"], out_d["pred_masks"]] + [t for a in out_d["aux_outputs"] for t in a.values()] + out_d["middles_attn_mask"]
        targets = model.prepare_targets([i for b in batch for i in b["instances"]], images)
        losses = model.criterion(out_d, targets)
        total = torch.stack([v * model.criterion.weight_dict[k] for k, v in losses.items()]).sum()
        tm = T()
        with tm.section("criterion bwd"):
            g_out = torch.autograd.grad(total, leaves_d, allow_unused=True)
        g_out = [g if g is not None else torch.zeros_like(l) for g, l in zip(g_out, leaves_d)]
        with tm.section("decoder bwd"):
            g = torch.autograd.grad(leaves, ms_d + [a256_d, fv_d] + P(head.predictor), g_out, allow_unused=True, retain_graph=bool(os.environ.get("PROF_STAGE")))
        g_ms, g_a256, g_fv = g[:3], g[3], g[4]
        with tm.section("fusion+mlp bwd"):
            g2 = torch.autograd.grad([fused["visual"]["res2"], a256], [mf_d] + P(head.fusion_module) + P(head.audio_transformation), [g_fv, g_a256], allow_unused=True)
        if os.environ.get("PROF_STAGE") == "pd" and it == 3:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                g3 = torch.autograd.grad([mf] + list(ms), list(feats_d.values()) + P(head.pixel_decoder), [g2[0]] + list(g_ms), allow_unused=True, retain_graph=True)
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=60))
        if os.environ.get("PROF_STAGE") == "dec" and it == 3:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                g = torch.autograd.grad(leaves, ms_d + [a256_d, fv_d] + P(head.predictor), g_out, allow_unused=True, retain_graph=True)
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=60))
        with tm.section("pixel decoder bwd"):
            g3 = torch.autograd.grad([mf] + list(ms), list(feats_d.values()) + P(head.pixel_decoder), [g2[0]] + list(g_ms), allow_unused=True)
        with tm.section("semmix+backbones bwd"):
            g4 = torch.autograd.grad(list(feats.values()), P(model.backbone) + P(model.pre_sam_backbone) + P(model.scale_factor_module), list(g3[:4]), allow_unused=True)
        if it >= 1:
            print("BWD | " + " | ".join(f"{k} {v * 1e3:.1f}" for k, v in tm.t.items()))
