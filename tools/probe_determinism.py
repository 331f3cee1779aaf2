#!/usr/bin/env python3
"""eager vs eager and eager vs replayed training step, per parameter, under module-constant overrides:
    python tools/probe_determinism.py [module.CONST=value ...]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import graph_compare as GC  # noqa: E402

import combo_avs_amd  # noqa: E402,F401

for a in sys.argv[1:]:
    k, v = a.split("=")
    mod, name = k.rsplit(".", 1)
    setattr(importlib.import_module(mod), name, eval(v))
model, opt, batches, _ = GC.build("r50")
from combo_avs_amd.trainer import GraphedTrainStep, train_step  # noqa: E402

snap = opt.flat_param.clone()
runs = []
for _ in range(3):
    GC.reset(opt, snap)
    losses = train_step(model, opt, batches[0])
    runs.append(({k: float(v) for k, v in losses.items()}, opt.flat_grad.clone()))
graphed = GraphedTrainStep(model, opt)
GC.reset(opt, snap)
gl = graphed(batches[0])
runs.append(({k: float(v) for k, v in gl.items()}, opt.flat_grad.clone()))
for name, i, j in (("eager2-eager1", 1, 0), ("eager3-eager1", 2, 0), ("graph-eager1", 3, 0)):
    rep = GC.per_parameter(opt, runs[i][1], runs[j][1])
    head = [r for r in rep if not r[0].startswith(("backbone.", "pre_sam_backbone."))]
    lib = [r for r in rep if r[0].startswith(("backbone.", "pre_sam_backbone."))]
    dl = max(abs(runs[i][0][k] - runs[j][0][k]) / (abs(runs[j][0][k]) + 1e-12) for k in runs[j][0])
    print(name, "max rel loss diff %.2e" % dl)
    for label, rows in (("head", head), ("backbones", lib)):
        w = sorted(rows, key=lambda r: -r[3])[:4]
        print("   ", label, ", ".join(f"{r[0][-50:]} {r[3]:.1e}/{r[4]:.2f}" for r in w))
