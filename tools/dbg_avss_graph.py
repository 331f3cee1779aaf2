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
g = GraphedTrainStep(model, opt, pad_targets_to=4 if "pad" in mode else None)
padded, counts = (g._pad_instances(b1) if "pad" in mode else (b1, None))
flags = g._avss_flags(b1)
dev = torch.device("cuda")
model.avss_static_index = (torch.tensor([i for i, v in enumerate(flags[0]) if v], device=dev), torch.tensor([i for i, v in enumerate(flags[1]) if v], device=dev))
crit = model.criterion
if counts is not None:
    crit.padded_counts = torch.tensor(counts, dtype=torch.int32, device=dev)
crit.num_masks_override = torch.tensor([float(sum(counts) if counts else 25)], device=dev)
from combo_avs_amd.ops.linear import grouped_presplit

def fwd():
    with grouped_presplit():
        l = model(padded)
        tot = getattr(l, "total", None)
        if tot is None:
            tot = torch.stack(list(l.values())).sum()
        if "bwd" in mode:
            opt.backward(tot)
    return tot.detach()

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        if "nograd" in mode:
            with torch.no_grad():
                fwd()
        else:
            fwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warm-up done", flush=True)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode="thread_local"):
    if "nograd" in mode:
        with torch.no_grad():
            out = fwd()
    else:
        out = fwd()
print("captured", flush=True)
for i in range(3):
    gr.replay()
    torch.cuda.synchronize()
    print("replay", i, float(out), flush=True)
