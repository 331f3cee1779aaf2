"""FlatAdamW against torch.optim.AdamW built the way the reference builds it (train_net.py:170-226: one group per parameter,
backbone lr x 0.1, no decay on norms / embeddings, full-model grad-norm clip): parameters without a gradient are skipped
like torch does, and the optimizer state moves in both directions through torch's state_dict layout (resume)."""
import copy

import torch
from torch import nn


def _model():
    torch.manual_seed(1)
    return nn.ModuleDict({
        "backbone": nn.Sequential(nn.Linear(9, 7), nn.LayerNorm(7)),
        "head": nn.Sequential(nn.Linear(7, 5), nn.GroupNorm(1, 5)),
        "unused": nn.Linear(5, 3),           # never reached by the loss: grad None in torch, skipped by AdamW
        "emb": nn.Embedding(4, 5)})


def _loss(m, x):
    y = m["head"](m["backbone"](x))
    return (y * m["emb"].weight[:y.shape[0]].sum(0)).pow(2).sum() * 20


def _torch_opt(m):
    from combo_avs_amd.trainer import param_groups
    groups = [{"params": [p], "lr": lr, "weight_decay": wd} for p, _, lr, wd in param_groups(m, 1e-2, 0.05)]
    return torch.optim.AdamW(groups, 1e-2), [p for g in groups for p in g["params"]]


def _step_ref(m, ref, plist, x):
    ref.zero_grad(set_to_none=True)
    _loss(m, x).backward()
    torch.nn.utils.clip_grad_norm_([p for p in plist if p.grad is not None], 0.01)
    ref.step()


def _step_flat(m, opt, x):
    opt.backward(_loss(m, x))
    opt.all_reduce_grads()
    opt.step()


def _same(a, b, tol=1e-6):
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=tol, msg=n)


def test_unused_parameters_are_skipped_and_state_round_trips_through_the_torch_layout():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.trainer import FlatAdamW
    a, b = _model(), _model()
    opt = FlatAdamW(a, base_lr=1e-2, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01)
    ref, plist = _torch_opt(b)
    w_unused = a["unused"].weight.detach().clone()
    xs = [torch.randn(3, 9) for _ in range(6)]
    for x in xs[:2]:
        _step_flat(a, opt, x)
        _step_ref(b, ref, plist, x)
    _same(a, b)
    assert torch.equal(a["unused"].weight, w_unused)  # no weight decay, no drift (torch skips grad-None parameters)
    assert len(opt.unused) == 2 and len(opt._active_segments()) > len(opt.segments) - 1

    # torch -> flat: resume from the reference optimizer's checkpoint
    a2 = copy.deepcopy(b)
    opt2 = FlatAdamW(a2, base_lr=1e-2, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.step_count == 2
    # flat -> torch: the state dict loads into torch.optim.AdamW
    b2 = copy.deepcopy(a)
    ref2, plist2 = _torch_opt(b2)
    ref2.load_state_dict(opt.state_dict())
    for x in xs[2:4]:
        _step_flat(a, opt, x)
        _step_ref(b, ref, plist, x)
        _step_flat(a2, opt2, x)
        _step_ref(b2, ref2, plist2, x)
    _same(a, b)
    _same(a2, b)
    _same(b2, b)
    # flat -> flat
    a3 = copy.deepcopy(a)
    opt3 = FlatAdamW(a3, base_lr=1e-2, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01)
    opt3.load_state_dict(opt.state_dict())
    for x in xs[4:]:
        _step_flat(a, opt, x)
        _step_flat(a3, opt3, x)
    _same(a, a3, tol=0.0)
