"""Synthetic inputs/targets shared by gen_golden.py (reference side) and the tests (oracle / HIP side)."""
import torch

import synth

HEAD_SEED = 0
BT = 5


def head_inputs(bt=BT, hw=56):
    feats = {}
    for i, c in enumerate((256, 512, 1024, 2048)):
        s = hw // (2 ** i)
        feats[f"res{i + 2}"] = synth.synth_tensor(f"feat.res{i + 2}", (bt, c, s, s), HEAD_SEED)
    audio = synth.synth_tensor("feat.audio", (bt, 1, 128), HEAD_SEED).abs()  # VGGish ends with a ReLU
    return feats, audio


def make_targets(mode, bt=BT, size=224):
    """Synthetic GT: per GT frame two complementary blob masks, classes [0, 1] (S4/MS3: K=2)."""
    n = {"s4": bt // 5, "all": bt}[mode]
    targets = []
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    for i in range(n):
        g = synth.rng_of(f"target.{mode}.{i}", HEAD_SEED)
        cx, cy, r = g.uniform(60, 160), g.uniform(60, 160), g.uniform(25, 70)
        blob = ((xx - cx) ** 2 + (yy - cy) ** 2) < r * r
        masks = torch.stack([~blob, blob]).to(torch.bool)
        targets.append({"labels": torch.tensor([0, 1], dtype=torch.int64), "masks": masks})
    return targets
