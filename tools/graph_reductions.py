#!/usr/bin/env python3
"""Which ATen reductions of one training step may split an output over several workgroups (Reduce.cuh's global reduce: staging
buffer + semaphores + a memset per launch)?  Those are the launches tools/graph_reduce_repro.py shows to be unreliable when
replayed from a hipGraph on this stack, so the captured step must not contain any.  A TorchDispatchMode lists every aten
reduction (and every convolution_backward that asks for the bias gradient, which reduces inside the C++ op) with its reduction
length = input numel / output numel; candidates are the ones above `MIN_LEN` (default 2048).

usage: python tools/graph_reductions.py [bench config name, default r50_s4]"""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import combo_avs_amd  # noqa
import bench
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW, train_step

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "r50_s4"
wl = bench.WORKLOADS[name]
cfg = combo_cfg(os.path.join(ROOT, "configs", wl["yaml"]), opts=wl.get("opts", ()))
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
if wl["dtype"] == "bf16":
    model.backbone_dtype = torch.bfloat16
opt = FlatAdamW(model, clip_value=cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE)
batch = bench.synth_batch(wl["clips"], wl["T"], wl["HW"], wl["HW"], dev, seed=100, K=wl["K"], gt=wl["gt"], avss=wl["avss"])
for _ in range(2):
    train_step(model, opt, batch)
torch.cuda.synchronize()
RED = ("sum", "mean", "amax", "amin", "max", "min", "norm", "linalg_vector_norm", "var_mean", "var", "std", "prod", "logsumexp",
       "any", "all", "argmax", "argmin", "nansum", "count_nonzero")
MIN_LEN = int(os.environ.get("MIN_LEN", "2048"))
agg = collections.defaultdict(int)


def site_of():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(ROOT) and "/tools/" not in fr.filename:
            return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
    return "autograd engine"


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        op = func.__name__.split(".")[0]
        if op in RED and args and torch.is_tensor(args[0]) and args[0].is_cuda:
            o = out[0] if isinstance(out, (tuple, list)) else out
            n_in, n_out = args[0].numel(), max(o.numel(), 1)
            if n_in // n_out >= MIN_LEN:
                agg[(op, tuple(args[0].shape), str(args[0].dtype)[6:], tuple(o.shape), site_of())] += 1
        elif op == "convolution_backward" and args[-1][2]:
            dy = args[0]
            n_out = dy.shape[1]
            if dy.numel() // n_out >= MIN_LEN:
                agg[("convolution_backward(bias)", tuple(dy.shape), str(dy.dtype)[6:], (n_out,), site_of())] += 1
        return out


with Spy():
    train_step(model, opt, batch)
torch.cuda.synchronize()
print(f"config {name}: reductions with >= {MIN_LEN} inputs per output in one training step")
in_graph = 0
for (op, shp, dt, oshp, site), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    eager = site.startswith("combo-avs_amd/trainer.py")  # the optimizer step runs eagerly after the replayed graph
    in_graph += 0 if eager else n
    print(f"  {n:4d} x {op:28s} {dt:9s} {str(shp):28s} -> {str(oshp):16s} {site}{' (optimizer step: not captured)' if eager else ''}")
print("candidate launches inside the captured step:", in_graph)
