"""Synthetic inputs/targets shared by gen_golden.py (reference side) and the tests (oracle / HIP side)."""
import torch

import synth

HEAD_SEED = 0
BT = 5


def head_inputs(bt=BT, hw=56, channels=(256, 512, 1024, 2048), tag="feat"):
    feats = {}
    for i, c in enumerate(channels):
        s = hw // (2 ** i)
        feats[f"res{i + 2}"] = synth.synth_tensor(f"{tag}.res{i + 2}", (bt, c, s, s), HEAD_SEED)
    audio = synth.synth_tensor(f"{tag}.audio", (bt, 1, 128), HEAD_SEED).abs()  # VGGish ends with a ReLU
    return feats, audio


def make_targets_k(frames, size, K, tag):
    """Synthetic AVSS-style GT (round 6, BASELINE configs[3]): per annotated frame 1 - 4 of the K classes (sorted, as np.unique
    returns them in avss_semantic_dataset_mapper.py:218-231), one blob mask each (blobs may overlap or miss each other: the
    criterion treats the masks as independent binary targets)."""
    targets = []
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    for i in range(frames):
        g = synth.rng_of(f"target.{tag}.{i}", HEAD_SEED)
        n = int(g.integers(1, 5))
        cls = sorted(int(c) for c in g.choice(K, size=n, replace=False))
        masks = []
        for _ in range(n):
            cx, cy, r = g.uniform(0.25, 0.75) * size, g.uniform(0.25, 0.75) * size, g.uniform(0.1, 0.3) * size
            masks.append(((xx - cx) ** 2 + (yy - cy) ** 2) < r * r)
        targets.append({"labels": torch.tensor(cls, dtype=torch.int64), "masks": torch.stack(masks).to(torch.bool)})
    return targets


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
