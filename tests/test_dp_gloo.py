"""CPU, world_size 2, gloo: the data-parallel host logic (flat gradient buffer, ONE all-reduce, averaging, global-norm
clip, grouped AdamW) gives the same parameters as a single process that sees the whole batch and uses the optimiser
the reference builds (train_net.py:147-226).  Also covers the criterion's `num_masks` all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(3)
    return torch.nn.ModuleDict({
        "backbone": torch.nn.Sequential(torch.nn.Linear(12, 10), torch.nn.LayerNorm(10)),
        "sem_seg_head": torch.nn.Sequential(torch.nn.Linear(10, 4)),
        "emb": torch.nn.Embedding(3, 4)})


def _loss(m, x, y):
    out = m["sem_seg_head"](m["backbone"](x)) + m["emb"].weight[0]
    return ((out - y) ** 2).mean() * 100


def _worker(rank, world, port, ret, comm_dtype=torch.float32, cut=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.modeling.criterion import SetCriterion
    from combo_avs_amd.trainer import FlatAdamW
    m = _model()
    opt = FlatAdamW(m, base_lr=1e-2, weight_decay=0.05, backbone_multiplier=0.1, clip_value=0.01, grad_comm_dtype=comm_dtype,
                    early=(lambda n: not n.startswith("backbone")) if cut else None)
    if cut:
        assert 0 < opt.n_early < len(opt.params) and 0 < opt.split < opt.numel
    g = torch.Generator().manual_seed(7)
    for it in range(4):
        x, y = torch.randn(8, 12, generator=g), torch.randn(8, 4, generator=g)
        xs, ys = x[rank::world], y[rank::world]  # clip i -> rank i mod world
        if cut:  # the backward pass cut at the head's input: early region reduced first (asynchronously on a GPU), then the rest
            feat = m["backbone"](xs)
            out = m["sem_seg_head"](feat) + m["emb"].weight[0]
            loss = ((out - ys) ** 2).mean() * 100
            cg = opt.backward_early(loss, [feat])
            opt.all_reduce_grads("early", overlap=True)
            opt.backward_late(cg)
            opt.all_reduce_grads("late")
            opt.wait_comm()
        else:
            opt.backward(_loss(m, xs, ys))
            opt.all_reduce_grads()
        opt.step()
    # num_masks: sum over ranks / world, clamped at 1 (criterion.py:261-265)
    crit = SetCriterion(2, matcher=None, weight_dict={}, eos_coef=0.1, losses=[], num_points=4, oversample_ratio=3.0,
                        importance_sample_ratio=0.75)
    nm = crit._num_masks([{"labels": torch.zeros(1 + 2 * rank)}], torch.device("cpu"))
    if rank == 0:
        ret["params"] = {k: v.detach().clone() for k, v in m.state_dict().items()}
        ret["num_masks"] = float(nm)
        ret["numel"] = opt.numel
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_reference_optimizer():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.trainer import param_groups
    m = _model()
    groups = [{"params": [p], "lr": lr, "weight_decay": wd} for p, _, lr, wd in param_groups(m, 1e-2, 0.05)]
    ref = torch.optim.AdamW(groups, 1e-2)
    g = torch.Generator().manual_seed(7)
    for it in range(4):
        x, y = torch.randn(8, 12, generator=g), torch.randn(8, 4, generator=g)
        ref.zero_grad()
        # DDP semantics: mean over ranks of per-rank mean losses == mean over the whole batch (equal shards)
        _loss(m, x, y).backward()
        torch.nn.utils.clip_grad_norm_([p for gr in groups for p in gr["params"]], 0.01)
        ref.step()
    for k, v in m.state_dict().items():
        torch.testing.assert_close(ret["params"][k], v, rtol=1e-5, atol=1e-7, msg=k)
    assert ret["num_masks"] == 2.0  # (1 + 3) / 2


def test_cut_backward_with_two_region_all_reduce_equals_the_single_collective():
    """The data-parallel overlap path (trainer.train_step / GraphedTrainStep with world_size > 1): the backward pass cut at the
    head's input, the head's region of the flat gradient buffer all-reduced first, the backbone's region after the second
    half - must give exactly the parameters of the one-collective path."""
    res = []
    for cut in (False, True):
        port = _free_port()
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(2, port, ret, torch.float32, cut), nprocs=2, join=True)
        res.append(dict(ret["params"]))
    for k in res[0]:
        torch.testing.assert_close(res[0][k], res[1][k], rtol=1e-6, atol=1e-8, msg=k)


def test_bf16_gradient_transport_stays_close_to_fp32():
    """SURVEY 8(f) rank 1 option: the flat gradient is all-reduced in bf16 (fp32 master weights and optimiser state)."""
    res = []
    for dt in (torch.float32, torch.bfloat16):
        port = _free_port()
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(2, port, ret, dt), nprocs=2, join=True)
        res.append(dict(ret["params"]))
    moved = 0.0
    for k in res[0]:
        d = (res[0][k] - res[1][k]).abs().max()
        assert d <= 3e-2 * 4 * 1e-2 + 1e-6, (k, float(d))  # 4 AdamW steps of at most lr = 1e-2 each: a few % of the motion
        moved = max(moved, float(d))
    assert moved > 0  # the bf16 path really ran


def test_param_group_rules():
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.trainer import param_groups
    m = _model()
    rules = {n: (lr, wd) for _, n, lr, wd in param_groups(m, 1e-4, 0.05)}
    assert rules["backbone.0.weight"] == (1e-5, 0.05)       # "backbone" in module name -> lr x 0.1
    assert rules["backbone.1.weight"] == (1e-5, 0.0)        # LayerNorm -> WEIGHT_DECAY_NORM = 0
    assert rules["sem_seg_head.0.bias"] == (1e-4, 0.05)
    assert rules["emb.weight"] == (1e-4, 0.0)               # nn.Embedding -> WEIGHT_DECAY_EMBED = 0
