#!/usr/bin/env python3
"""Which parameters' gradients does FlatAdamW receive in place (as their flat-buffer views), which through the concatenation?"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import graph_compare as GC
model, opt, batches, state = GC.build("r50")
orig = opt._collect
stat = collections.defaultdict(lambda: [0, 0])
def spy(grads, lo, hi):
    for g, (p, name, _, _), view in zip(grads, opt.entries[lo:hi], opt.grad_views[lo:hi]):
        kind = "none" if g is None else ("inplace" if (g.data_ptr() == view.data_ptr() and g.is_contiguous()) else "copied")
        key = (kind, name.split(".")[0], tuple(p.shape[2:]) if p.dim() == 4 else p.dim())
        stat[key][0] += 1; stat[key][1] += p.numel()
        if kind == "copied" and p.dim() == 4 and p.numel() > 1e6 and stat[key][0] <= 2:
            print("copied:", name, tuple(p.shape), "grad ptr", g.data_ptr(), "view ptr", view.data_ptr(), g.is_contiguous(), g.stride())
    return orig(grads, lo, hi)
opt._collect = spy
total = sum(model(batches[0]).values())
opt.backward(total)
torch.cuda.synchronize()
for k, v in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(k, v[0], f"{v[1] * 4 / 1e6:.1f} MB")
