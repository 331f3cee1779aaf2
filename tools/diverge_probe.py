#!/usr/bin/env python3
"""Runs a bench workload eagerly for N steps and prints the weighted total loss per step, the first non-finite loss entry and the
LSAP status word (a diverged step): `python tools/diverge_probe.py pvt_ms3_t10 10`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import combo_avs_amd  # noqa
from combo_avs_amd import combo_cfg
from combo_avs_amd.meta_arch import build_model
from combo_avs_amd.trainer import FlatAdamW, train_step

name, steps = sys.argv[1], int(sys.argv[2])
wl = bench.WORKLOADS[name]
cfg = combo_cfg(os.path.join(bench.ROOT, "configs", wl["yaml"]), opts=wl.get("opts", ()))
torch.manual_seed(0)
dev = torch.device("cuda")
model = build_model(cfg).to(dev).train()
if wl["dtype"] == "bf16":
    model.backbone_dtype = torch.bfloat16
opt = FlatAdamW(model, base_lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY, backbone_multiplier=cfg.SOLVER.BACKBONE_MULTIPLIER,
                clip_value=cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE,
                early=(lambda n: n.startswith("sem_seg_head.")) if os.environ.get("PROBE_EARLY") == "1" else None)
seeds = [100, 1000, 2000, 3000] if os.environ.get("PROBE_SEEDS") == "bench" else [100 + 1000 * i for i in range(4)]
batches = [bench.synth_batch(wl["clips"], wl["T"], wl["HW"], wl["HW"], dev, seed=sd, K=wl["K"], gt=wl["gt"], avss=wl["avss"]) for sd in seeds]
graphed = None
if os.environ.get("PROBE_GRAPH") == "1":  # the bench's own sequence: one eager step, then the captured graph
    from combo_avs_amd.trainer import GraphedTrainStep
    graphed = GraphedTrainStep(model, opt)
    if os.environ.get("PROBE_MIOPEN_BENCH", "0") == "1":
        torch.backends.cudnn.benchmark = True
if os.environ.get("PROBE_OWN_DB") == "1":  # bias gradients of the PVT linears WITHOUT ATen's multi-block reduction (semaphores +
    # a memset node per reduction in the captured graph): db = ones[1,M] @ dy
    import combo_avs_amd.backbone_pvt as BP
    import torch.nn.functional as F

    class _Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return F.linear(x, w, b)

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
            ones = torch.ones(1, dy2.shape[0], dtype=dy2.dtype, device=dy2.device)
            return (dy2 @ w).view_as(x), dy2.t() @ x2, (ones @ dy2).view(-1)

    def _linear(x, mod, wts):
        w = BP._p(mod.weight, wts)
        x = x if x.dtype == w.dtype else x.to(w.dtype)
        if mod.bias is None:
            return F.linear(x, w)
        return _Lin.apply(x, w, BP._p(mod.bias, wts))
    BP._linear = _linear

    class _AddBias(torch.autograd.Function):  # y[B,C,H,W] + b[C]; db through a GEMM with a row of ones
        @staticmethod
        def forward(ctx, y, b):
            return y + b.view(1, -1, 1, 1)

        @staticmethod
        def backward(ctx, dy):
            d2 = dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1])
            ones = torch.ones(1, d2.shape[0], dtype=d2.dtype, device=d2.device)
            return dy, (ones @ d2).view(-1)

    def _conv(x, mod, wts):
        w = BP._p(mod.weight, wts)
        y = F.conv2d(x if x.dtype == w.dtype else x.to(w.dtype), w, None, mod.stride, mod.padding, mod.dilation, mod.groups)
        return y if mod.bias is None else _AddBias.apply(y, BP._p(mod.bias, wts))
    if os.environ.get("PROBE_OWN_CONV_DB", "1") == "1":
        BP._conv = _conv
ts_buf = None
order = [int(x) for x in os.environ.get("PROBE_ORDER", "").split(",") if x] or [i % 4 for i in range(steps)]
steps = len(order)
named = None
for it in range(steps):
    if it == 1 and os.environ.get("PROBE_TS") == "1":  # the bench's device-side timing slots
        from combo_avs_amd import _lib as _clib
        ts_buf = torch.zeros(4096, 256, dtype=torch.int64, device=dev)
        ts_buf[:, 0::16] = -1
        _clib.check(_clib.lib().combo_timing_set_buffer(ts_buf.data_ptr(), 4096), "combo_timing_set_buffer")
    if ts_buf is not None and it == int(os.environ.get("PROBE_ZERO_AT", "-1")):  # the bench's reset of the launch counters
        kind = os.environ.get("PROBE_ZERO_KIND", "ts")
        if kind != "nosync":
            torch.cuda.synchronize()
        if kind in ("ts", "nosync"):
            ts_buf[:, 2:4] = 0
        elif kind == "other":
            junk = torch.ones(4096, 256, dtype=torch.int64, device=dev)
            junk[:, 2:4] = 0
        elif kind == "memset":
            ts_buf.view(-1)[: 16].zero_()
        elif kind == "sleep":
            import time
            time.sleep(1.0)
        if kind != "nosync":
            torch.cuda.synchronize()
    losses = train_step(model, opt, batches[order[it]]) if (graphed is None or it == 0) else graphed(batches[order[it]])
    if ts_buf is not None:
        _clib.lib().combo_timing_fold(_clib.current_stream())
    tot = sum(float(v) for v in losses.values())
    bad = [k for k, v in losses.items() if not torch.isfinite(v)]
    gn = float(torch.linalg.vector_norm(opt.flat_grad))
    print(f"step {it}: total {tot:.4f} grad-norm {gn:.4g} params-finite {bool(torch.isfinite(opt.flat_param).all())} non-finite {bad[:3]}", flush=True)
    if not (gn < 1e5) or os.environ.get("PROBE_TOP") == "1":
        nm = {p.data_ptr(): n for n, p in model.named_parameters()}
        big = sorted(((float(v.abs().max()), nm.get(p.data_ptr(), "?")) for p, v in zip(opt.params, opt.grad_views)), reverse=True)
        print("  largest |grad| per tensor:", [(f"{a:.3g}", n) for a, n in big[:int(os.environ.get("PROBE_TOPN", "40"))]])
    if not torch.isfinite(opt.flat_grad).all() and named is None:
        names = {p.data_ptr(): n for n, p in model.named_parameters()}
        gv = [(names.get(p.data_ptr(), "?"), v) for p, v in zip(opt.params, opt.grad_views)]
        named = [n for n, v in gv if not torch.isfinite(v).all()]
        print(f"  first non-finite gradients in {len(named)} tensors: {named[:12]}")
        fin = [n for n, v in gv if torch.isfinite(v).all()]
        print(f"  finite in {len(fin)} tensors: {fin[:12]}")
    try:
        model.criterion.matcher.check_status()
    except ValueError as e:
        print("  LSAP status:", e)
