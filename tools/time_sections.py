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
