#!/usr/bin/env python3
"""Where the ATen elementwise / copy / reduction launches of one eager fp32 training step at BASELINE configs[1] come from:
a TorchDispatchMode sees every aten op with its argument shapes; forward ops carry the Python stack (innermost frame inside
this repository), ops issued by the C++ autograd engine do not (tagged "autograd engine").  Bytes = numel of the output x 4."""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import combo_avs_amd  # noqa
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW, train_step
from bench import synth_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "r50_s4"]  # usage: tools/aten_sites.py [bench config]
cfg = combo_cfg(os.path.join(ROOT, "configs", wl["yaml"]), opts=wl.get("opts", ()))
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
if wl["dtype"] == "bf16":
    model.backbone_dtype = torch.bfloat16
opt = FlatAdamW(model)
batch = synth_batch(wl["clips"], wl["T"], wl["HW"], wl["HW"], dev, seed=1, K=wl["K"], gt=wl["gt"], avss=wl["avss"])
for _ in range(2):
    train_step(model, opt, batch)
torch.cuda.synchronize()
WATCH = ("add", "add_", "copy_", "sum", "fill_", "zero_", "mul", "div", "cat", "clone", "_to_copy", "sub", "mul_", "index_select",
         "stack", "where", "neg", "mean", "zeros_like", "zeros", "contiguous", "index", "index_put_", "_foreach_copy_")
agg = collections.defaultdict(lambda: [0, 0])


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = out if torch.is_tensor(out) else (args[0] if args and torch.is_tensor(args[0]) else None)
            if t is not None and t.is_cuda:
                node = torch._C._current_autograd_node() if hasattr(torch._C, "_current_autograd_node") else None
                site = "autograd engine" + (f" ({node.name()})" if node is not None else "")
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "combo-avs_amd" in fr.filename and "trainer.py" not in fr.filename:
                        site = f"{fr.filename.split('combo-avs_amd/')[-1]}:{fr.lineno} {(fr.line or '')[:60]}"
                        break
                flat = [a for x in args for a in (x if isinstance(x, (list, tuple)) else [x])]
                shapes = str([tuple(a.shape) for a in flat if torch.is_tensor(a)][:3])[:60]
                k = (name, shapes, site)
                agg[k][0] += 1
                agg[k][1] += t.numel() * t.element_size()
        return out


with Spy():
    train_step(model, opt, batch)
torch.cuda.synchronize()
tot_n = sum(v[0] for v in agg.values())
tot_b = sum(v[1] for v in agg.values())
print(f"watched ATen ops: {tot_n} calls, {tot_b / 1e6:.0f} MB of outputs")
for (name, shapes, site), (n, b) in sorted(agg.items(), key=(lambda kv: -kv[1][0]) if os.environ.get("SORT") == "calls" else (lambda kv: -(kv[1][1] + 2e5 * kv[1][0])))[:int(os.environ.get("TOP", "70"))]:
    print(f"{n:4d}x {b / 1e6:8.1f} MB  {name:14s} {shapes:60s} {site}")
