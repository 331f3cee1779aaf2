import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import combo_avs_amd
from bench import synth_batch
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW, GraphedTrainStep, train_step
mode = sys.argv[1]
cfg = combo_cfg(os.path.join(ROOT, "configs/avs_ss/COMBO_PVTV2B5_bs8_90k.yaml"))
torch.manual_seed(0)
model = build_model(cfg).cuda().train()
model.backbone_dtype = torch.bfloat16
opt = FlatAdamW(model, base_lr=0.0, weight_decay=0.0, backbone_multiplier=0.1, clip_value=0.01)
b1 = synth_batch(1, 10, 224, 224, "cuda", seed=5, K=71, gt="all", avss=True)
b2 = synth_batch(1, 10, 224, 224, "cuda", seed=6, K=71, gt="all", avss=True)
g = GraphedTrainStep(model, opt, pad_targets_to=4)
second = b1
if mode == "images":
    second = [dict(b1[0], images=b2[0]["images"], pre_masks=b2[0]["pre_masks"], audio_log_mel=b2[0]["audio_log_mel"])]
elif mode == "instances":
    second = [dict(b1[0], instances=b2[0]["instances"])]
elif mode == "masks_only":  # same counts / classes as b1, other mask contents
    inst = []
    for i1 in b1[0]["instances"]:
        inst.append({"gt_classes": i1["gt_classes"], "gt_masks": torch.roll(i1["gt_masks"], 17, -1)})
    second = [dict(b1[0], instances=inst)]
elif mode == "classes_only":
    inst = []
    for i1 in b1[0]["instances"]:
        inst.append({"gt_classes": (i1["gt_classes"] + 3) % 71, "gt_masks": i1["gt_masks"]})
    second = [dict(b1[0], instances=inst)]
for name, b in (("first", b1), ("second", second), ("second", second)):
    l = g(b)
    torch.cuda.synchronize()
    print(mode, name, "ok, graphs", len(g.graphs), float(sum(l.values())), flush=True)
