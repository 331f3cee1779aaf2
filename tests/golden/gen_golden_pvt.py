#!/usr/bin/env python3
"""Golden vectors for the PVTv2-B5 backbone (SURVEY 8(b) registry surface / 8(f) rank 2): imports the reference's
models/modeling/backbone/pvtv2.py in THIS container (stand-ins for the un-installed timm / detectron2 names only; nothing
is copied), loads name-seeded synthetic weights (tests/golden/synth.py), runs eval-mode forward + backward on a seeded
input and stores digests.  Run from the repo root:  python tests/golden/gen_golden_pvt.py"""
import importlib
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402
import synth  # noqa: E402


class _EvalDropPath(nn.Module):
    """timm DropPath stand-in: identity in eval mode (the only mode the golden vectors use)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not self.training or self.drop_prob == 0.0, "golden vectors are generated in eval mode"
        return x


def main():
    refshim.install()
    tl = sys.modules["timm.models.layers"]
    tl.DropPath = _EvalDropPath
    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tl.trunc_normal_ = lambda t, std=1.0, **k: nn.init.trunc_normal_(t, std=std)
    refshim._mod("timm.models.vision_transformer", _cfg=lambda **k: dict(k))
    reg = refshim._Registry("BACKBONE")
    refshim._mod("detectron2.modeling.backbone", Backbone=nn.Module)
    refshim._mod("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=reg)
    refshim._mod("detectron2.modeling.backbone.fpn", FPN=object, LastLevelMaxPool=object, LastLevelP6P7=object)
    pv = importlib.import_module("models.modeling.backbone.pvtv2")
    cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(PVT=types.SimpleNamespace(OUT_FEATURES=["res2", "res3", "res4", "res5"])))
    torch.manual_seed(0)
    model = pv.build_pvtv2_b5_backbone(cfg, None).eval()
    spec = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(synth.synth_state_dict(spec, seed=0))
    x = synth.synth_tensor("pvt.x", (2, 3, 64, 64), 0)
    out = model(x)
    names = ["res2", "res3", "res4", "res5"]
    z = {"spec_names": np.array([n for n, _ in spec]), "spec_shapes": np.array([",".join(map(str, s)) for _, s in spec]),
         "out_shapes": np.array([",".join(map(str, out[n].shape)) for n in names])}
    for n in names:
        synth.pack(f"out.{n}", synth.digest(out[n], f"pvt.out.{n}"), z)
    loss = sum((out[n] * synth.synth_tensor(f"pvt.g.{n}", tuple(out[n].shape), 0)).sum() for n in names)
    probe = ["patch_embed1.proj.weight", "block1.0.attn.sr.weight", "block1.2.mlp.dwconv.dwconv.weight", "block2.3.attn.kv.weight",
             "block3.17.attn.q.bias", "block3.39.mlp.fc2.weight", "block4.1.attn.proj.weight", "norm4.weight"]
    params = dict(model.named_parameters())
    grads = torch.autograd.grad(loss, [params[p] for p in probe])
    z["probe"] = np.array(probe)
    for p, g in zip(probe, grads):
        synth.pack(f"grad.{p}", synth.digest(g, f"pvt.grad.{p}"), z)
    np.savez_compressed(os.path.join(HERE, "pvt.npz"), **z)
    print("wrote pvt.npz:", {n: tuple(out[n].shape) for n in names}, "params", sum(v.numel() for v in model.state_dict().values()))


if __name__ == "__main__":
    main()
