#!/usr/bin/env python3
"""Python call sites of real copies (contiguous / clone / copy_ / reshape that copies / cat / stack) in one eager fp32 training step
at BASELINE config 2 (bs = 8): the tensor methods are wrapped and the innermost frame inside this repository is counted, with the
bytes moved.  (ATen copies issued from inside the C++ autograd engine do not pass through here.)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW, train_step
from bench import synth_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_R50_bs8_90k.yaml"))
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
opt = FlatAdamW(model)
batch = synth_batch(8, 5, 224, 224, dev, 1)
train_step(model, opt, batch)
torch.cuda.synchronize()
sites = collections.defaultdict(lambda: [0, 0])


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "combo-avs_amd" in fr.filename and "copy_sites" not in fr.filename:
            return f"{fr.filename.split('combo-avs_amd/')[-1]}:{fr.lineno} {fr.line[:70]}"
    return "?"


def wrap(name, is_copy):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        try:
            if self.is_cuda and is_copy(self, out, a):
                s = sites[(name, site())]
                s[0] += 1
                s[1] += (out if torch.is_tensor(out) else self).numel() * (out if torch.is_tensor(out) else self).element_size()
        except Exception:
            pass
        return out
    setattr(torch.Tensor, name, f)


wrap("contiguous", lambda s, o, a: o.data_ptr() != s.data_ptr() or o is not s and not s.is_contiguous(*([] if not a else [])))
wrap("clone", lambda s, o, a: True)
wrap("copy_", lambda s, o, a: True)
wrap("reshape", lambda s, o, a: o.data_ptr() != s.data_ptr())
wrap("float", lambda s, o, a: o.dtype != s.dtype)
for fn in ("cat", "stack"):
    orig = getattr(torch, fn)

    def g(*a, _o=orig, _n=fn, **k):
        out = _o(*a, **k)
        if out.is_cuda:
            s = sites[(_n, site())]
            s[0] += 1
            s[1] += out.numel() * out.element_size()
        return out
    setattr(torch, fn, g)
train_step(model, opt, batch)
torch.cuda.synchronize()
for (name, where), (n, b) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{b / 1e6:9.1f} MB x{n:<4d} {name:10s} {where}")
