"""PVTv2-B5 backbone (combo_avs_amd/backbone_pvt.py, host PyTorch) against golden vectors generated
from the reference's pvtv2.py (tests/golden/gen_golden_pvt.py): same state-dict keys/shapes, same eval-mode features and
gradients on name-seeded synthetic weights; stochastic depth statistics; registry + config wiring."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth  # noqa: E402


@pytest.fixture(scope="module")
def pvt():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.config import combo_cfg
    from combo_avs_amd.registry import BACKBONE_REGISTRY
    cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_PVTV2B5_bs8_90k.yaml"))
    assert cfg.MODEL.BACKBONE.NAME == "build_pvtv2_b5_backbone"
    torch.manual_seed(0)
    model = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, None)
    z = np.load(os.path.join(ROOT, "tests/golden/pvt.npz"))
    return model, z


def test_state_dict_surface_equals_reference(pvt):
    model, z = pvt
    ref = dict(zip(z["spec_names"].tolist(), z["spec_shapes"].tolist()))
    mine = {k: ",".join(map(str, v.shape)) for k, v in model.state_dict().items()}
    assert mine == ref
    shapes = model.output_shape()
    assert [shapes[n].channels for n in ("res2", "res3", "res4", "res5")] == [64, 128, 320, 512]
    assert [shapes[n].stride for n in ("res2", "res3", "res4", "res5")] == [4, 8, 16, 32]


def test_eval_forward_and_gradients_match_reference(pvt):
    model, z = pvt
    spec = [(n, tuple(int(v) for v in s.split(","))) for n, s in zip(z["spec_names"].tolist(), z["spec_shapes"].tolist())]
    model.load_state_dict(synth.synth_state_dict(spec, seed=0))
    model.eval()
    x = synth.synth_tensor("pvt.x", (2, 3, 64, 64), 0)
    out = model(x)
    names = ["res2", "res3", "res4", "res5"]
    assert [",".join(map(str, out[n].shape)) for n in names] == z["out_shapes"].tolist()
    for n in names:
        synth.check_digest(out[n], synth.unpack(f"out.{n}", z), f"pvt.out.{n}", rtol=2e-4, atol=2e-4)
    loss = sum((out[n] * synth.synth_tensor(f"pvt.g.{n}", tuple(out[n].shape), 0)).sum() for n in names)
    params = dict(model.named_parameters())
    probe = z["probe"].tolist()
    grads = torch.autograd.grad(loss, [params[p] for p in probe])
    for p, g in zip(probe, grads):
        d = synth.unpack(f"grad.{p}", z)
        synth.check_digest(g, d, f"pvt.grad.{p}", rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(d["sample"]).max())))


def test_stochastic_depth_semantics():
    from combo_avs_amd.backbone_pvt import DropPath
    torch.manual_seed(0)
    dp = DropPath(0.25).train()
    x = torch.ones(4000, 3, 5)
    y = dp(x)
    kept = (y[:, 0, 0] != 0)
    assert abs(float(kept.float().mean()) - 0.75) < 0.03          # per-sample keep probability
    assert torch.allclose(y[kept], torch.full_like(y[kept], 1 / 0.75))  # rescaled by 1/(1-p), whole sample kept or dropped
    assert torch.equal(dp.eval()(x), x)


def test_training_mode_runs_and_drop_path_rates(pvt):
    model, _ = pvt
    rates = [getattr(b.drop_path, "p", 0.0) for i in range(4) for b in getattr(model, f"block{i + 1}")]
    assert len(rates) == 52 and rates[0] == 0.0 and abs(rates[-1] - 0.1) < 1e-9
    assert all(b >= a for a, b in zip(rates, rates[1:]))              # linear decay rule, pvtv2.py:262
    model.train()
    out = model(torch.randn(2, 3, 64, 64))
    assert all(torch.isfinite(v).all() for v in out.values())
