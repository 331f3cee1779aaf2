import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from bench import synth_batch
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW, GraphedTrainStep, train_step
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1]
cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_R50_bs8_90k.yaml"))
torch.manual_seed(0)
model = build_model(cfg).cuda().train()
if "nodrop" in mode:
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    for a in model.sem_seg_head.fusion_module.b_attn.attn_list:
        a.dropout = 0.0
if "bank" in mode:
    bank = torch.rand(40_000_000, generator=torch.Generator().manual_seed(5)).cuda()
    state = {"off": 0}
    def point_source(n, p):
        o = state["off"]
        state["off"] = o + n * p * 2
        return bank[o:o + n * p * 2].view(n, p, 2)
    model.criterion.point_source = point_source
    model.register_forward_pre_hook(lambda m, a: state.__setitem__("off", 0))
opt = FlatAdamW(model, base_lr=1e-4, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01) if "test" in mode else FlatAdamW(model)
batch = synth_batch(2, 5, 224, 224, "cuda", 11)
if "eager" in mode:
    train_step(model, opt, batch)
if "test" in mode:
    batch2 = synth_batch(2, 5, 224, 224, "cuda", 12)
    snap = opt.flat_param.clone()
    keep = []
    for b in (batch, batch2):
        if "reset" in mode:
            opt.flat_param.copy_(snap); opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_count = 0
        losses = train_step(model, opt, b)
        if "keep" in mode:
            keep.append(({k: float(v) for k, v in losses.items()}, opt.flat_grad.clone(), opt.flat_param.clone()))
    if "reset" in mode:
        opt.flat_param.copy_(snap); opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_count = 0
g = GraphedTrainStep(model, opt)
l = g(batch)
torch.cuda.synchronize()
print(mode, "OK", float(sum(l.values())))
