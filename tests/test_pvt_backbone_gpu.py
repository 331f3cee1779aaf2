"""GPU: the PVTv2-B5 backbone (combo_avs_amd/backbone_pvt.py) against the golden vectors generated from the
REFERENCE's pvtv2.py (tests/golden/gen_golden_pvt.py -> pvt.npz) - the comparison tests/test_pvt_backbone.py makes on the CPU,
on the device and in the two precisions the product runs the encoders in:

  fp32         eval-mode features and eight parameter gradients at the CPU test's bound (the library's fp32 kernels)
  bf16 recipe  the training path of the PVT workloads: bf16 autocast with the package's own kernels in it (pre-norm residual steps,
               bias + LayerNorm, depth-wise convolution, spatial-reduction attention), stochastic depth off - stated bound: relative
               L2 error of every feature map <= 2e-2 (bf16 operands through 52 blocks; measured below) and of the probed gradients <= 8e-2"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth  # noqa: E402

NAMES = ["res2", "res3", "res4", "res5"]


@pytest.fixture(scope="module")
def pvt():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.backbone_pvt import DropPath
    from combo_avs_amd.config import combo_cfg
    from combo_avs_amd.registry import BACKBONE_REGISTRY
    cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_PVTV2B5_bs8_90k.yaml"))
    model = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, None)
    z = np.load(os.path.join(ROOT, "tests/golden/pvt.npz"))
    spec = [(n, tuple(int(v) for v in s.split(","))) for n, s in zip(z["spec_names"].tolist(), z["spec_shapes"].tolist())]
    model.load_state_dict(synth.synth_state_dict(spec, seed=0))
    for m in model.modules():
        if isinstance(m, DropPath):
            m.p = 0.0
    return model.cuda(), z


def _loss(out):
    return sum((out[n].float() * synth.synth_tensor(f"pvt.g.{n}", tuple(out[n].shape), 0).cuda()).sum() for n in NAMES)


def test_fp32_features_and_gradients_match_the_reference(pvt):
    model, z = pvt
    model.eval()
    x = synth.synth_tensor("pvt.x", (2, 3, 64, 64), 0).cuda()
    out = model(x)
    assert [",".join(map(str, out[n].shape)) for n in NAMES] == z["out_shapes"].tolist()
    for n in NAMES:
        synth.check_digest(out[n].cpu(), synth.unpack(f"out.{n}", z), f"pvt.out.{n}", rtol=2e-4, atol=2e-4)
    params = dict(model.named_parameters())
    probe = z["probe"].tolist()
    grads = torch.autograd.grad(_loss(out), [params[p] for p in probe])
    for p, g in zip(probe, grads):
        d = synth.unpack(f"grad.{p}", z)
        synth.check_digest(g.cpu(), d, f"pvt.grad.{p}", rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(d["sample"]).max())))


def test_bf16_training_path_within_the_stated_bound(pvt):
    model, z = pvt
    model.train()  # (stochastic depth is off: DropPath.p = 0) - the pre-norm / own-kernel path needs grad mode
    x = synth.synth_tensor("pvt.x", (2, 3, 64, 64), 0).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
    worst = 0.0
    for n in NAMES:
        d = synth.unpack(f"out.{n}", z)
        idx = synth.digest_indices(out[n].numel(), 4096, f"pvt.out.{n}")
        got = out[n].float().contiguous().reshape(-1).cpu().numpy()[idx].astype(np.float64)
        ref = np.asarray(d["sample"]).astype(np.float64)
        rel = float(np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()))
        worst = max(worst, rel)
        assert rel <= 2e-2, (n, rel)
    params = dict(model.named_parameters())
    probe = z["probe"].tolist()
    grads = torch.autograd.grad(_loss(out), [params[p] for p in probe])
    gworst = 0.0
    for p, g in zip(probe, grads):
        d = synth.unpack(f"grad.{p}", z)
        idx = synth.digest_indices(g.numel(), 4096, f"pvt.grad.{p}")
        got = g.float().contiguous().reshape(-1).cpu().numpy()[idx].astype(np.float64)
        ref = np.asarray(d["sample"]).astype(np.float64)
        rel = float(np.sqrt(((got - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-300)))
        gworst = max(gworst, rel)
        assert rel <= 8e-2, (p, rel)
    print(f"[pvt bf16 recipe vs the reference's fp32 golden] worst relative L2: features {worst:.2e}, probed gradients {gworst:.2e}")
