import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import graph_compare as GC
from combo_avs_amd.trainer import GraphedTrainStep
model, opt, batches, state = GC.build("r50")
snap = opt.flat_param.clone()
for rep in range(4):
    GC.reset(opt, snap)
    old = [s[2] for s in opt.segments]
    for s in opt.segments: s[2] = s[2] * 20.0
    step = GraphedTrainStep(model, opt)
    tot = []
    for _ in range(40):
        tot.append(float(sum(step(batches[0]).values())))
    for s, o in zip(opt.segments, old): s[2] = o
    print(rep, tot[0], tot[10], tot[20], tot[-1], tot[-1] / tot[0], flush=True)
