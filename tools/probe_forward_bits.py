#!/usr/bin/env python3
"""which module's output differs bit-wise between two eager training forwards on the same input (first 40 in execution order)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import graph_compare as GC  # noqa: E402

import importlib  # noqa: E402

for a in sys.argv[1:]:
    k, v = a.split("=")
    mod, name = k.rsplit(".", 1)
    setattr(importlib.import_module(mod), name, eval(v))
model, opt, batches, _ = GC.build("r50")
from combo_avs_amd.trainer import train_step  # noqa: E402
snap = opt.flat_param.clone()
names = {m: n for n, m in model.named_modules()}
runs = []
for _ in range(2):
    rec = []

    def hook(m, inp, out, rec=rec):
        def flat(o):
            if torch.is_tensor(o):
                return [o]
            if isinstance(o, (list, tuple)):
                return [t for x in o for t in flat(x)]
            if isinstance(o, dict):
                return [t for x in o.values() for t in flat(x)]
            return []
        rec.append((names[m], [t.detach().float().double().sum().item() for t in flat(out)], [t.detach().clone() for t in flat(out)][:1]))
    hs = [m.register_forward_hook(hook) for m in model.modules()]
    GC.reset(opt, snap)
    losses = train_step(model, opt, batches[0])  # a whole step: the second run sees the allocator state the first one left
    for h in hs:
        h.remove()
    runs.append((rec, {k: float(v) for k, v in losses.items()}))
n = 0
for (na, sa, ta), (nb, sb, tb) in zip(runs[0][0], runs[1][0]):
    assert na == nb
    if ta and tb and ta[0].shape == tb[0].shape and not torch.equal(ta[0], tb[0]):
        d = (ta[0].float() - tb[0].float()).abs().max().item()
        print(f"DIFF {na or '<model>'}: max abs {d:.3e} of range {tb[0].float().abs().max().item():.3e}")
        n += 1
        if n >= 60:
            break
print("modules compared", len(runs[0][0]), "differing shown", n)
print("loss diffs", {k: runs[0][1][k] - runs[1][1][k] for k in list(runs[0][1])[:6]})
