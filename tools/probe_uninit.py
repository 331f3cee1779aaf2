#!/usr/bin/env python3
"""one eager training step with every torch.empty filled with NaN (torch.utils.deterministic.fill_uninitialized_memory): a kernel
that reads memory nobody wrote shows up as the first module with a NaN output / the parameters with a NaN gradient"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import graph_compare as GC  # noqa: E402

warnings.filterwarnings("ignore")
model, opt, batches, _ = GC.build("r50")
from combo_avs_amd.trainer import train_step  # noqa: E402

train_step(model, opt, batches[0])  # warm-up without the fill (library find steps)
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
names = {m: n for n, m in model.named_modules()}
bad = []


def hook(m, inp, out):
    def flat(o):
        if torch.is_tensor(o):
            return [o]
        if isinstance(o, (list, tuple)):
            return [t for x in o for t in flat(x)]
        if isinstance(o, dict):
            return [t for x in o.values() for t in flat(x)]
        return []
    for t in flat(out):
        if t.is_floating_point() and not torch.isfinite(t).all():
            bad.append(names[m])
            break


hs = [m.register_forward_hook(hook) for m in model.modules()]
losses = train_step(model, opt, batches[0])
for h in hs:
    h.remove()
print("modules with non-finite outputs (execution order):", bad[:12])
print("non-finite losses:", [k for k, v in losses.items() if not torch.isfinite(torch.as_tensor(float(v)))][:10])
nan_params = [name for (p, name, _, _), off in zip(opt.entries, opt.offsets) if not torch.isfinite(opt.flat_grad[off:off + p.numel()]).all()]
print("parameters with non-finite gradients:", len(nan_params), nan_params[:12])
