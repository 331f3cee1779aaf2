#!/usr/bin/env python3
"""Which ops of a training step call hipMemsetAsync (each call becomes a memset node of the captured hipGraph; see
tools/graph_reduce_repro.py for why the step should hold none)?  Run under rocprofv3 with the HIP API and marker traces:

  cd /tmp && rocprofv3 --hip-trace --marker-trace --output-format csv -d /tmp/ms -o t -- python3 $GRAFT_REPO_ROOT/tools/memset_sites.py run r50_s4
  python3 tools/memset_sites.py parse /tmp/ms

`run` executes two plain eager steps, then one step inside torch.autograd.profiler.emit_nvtx() (one roctx range per aten op and
per autograd node); `parse` reports, for every hipMemsetAsync / hipMemcpyAsync issued inside that step, the innermost ranges
around it."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name):
    import torch
    import combo_avs_amd  # noqa: F401
    import bench
    from combo_avs_amd import combo_cfg
    from combo_avs_amd.meta_arch import build_model
    from combo_avs_amd.trainer import FlatAdamW
    wl = bench.WORKLOADS[name]
    cfg = combo_cfg(os.path.join(ROOT, "configs", wl["yaml"]), opts=wl.get("opts", ()))
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = build_model(cfg).to(dev).train()
    if wl["dtype"] == "bf16":
        model.backbone_dtype = torch.bfloat16
    opt = FlatAdamW(model, clip_value=cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE)
    batch = bench.synth_batch(wl["clips"], wl["T"], wl["HW"], wl["HW"], dev, seed=100, K=wl["K"], gt=wl["gt"], avss=wl["avss"])

    def fwd_bwd():
        losses = model(batch)
        opt.backward(torch.stack(list(losses.values())).sum())
    for _ in range(2):
        fwd_bwd()
    torch.cuda.synchronize()
    from torch.utils._python_dispatch import TorchDispatchMode

    class Shapes(TorchDispatchMode):  # one roctx range per aten op with its tensor shapes (emit_nvtx(record_shapes=True) chokes
        # on the 64-bit seeds some of this package's autograd Functions carry)
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            shp = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            torch.cuda.nvtx.range_push(f"SHAPES {func.__name__} {shp} {[a for a in args if isinstance(a, (list, tuple)) and len(a) < 5 and all(isinstance(v, (int, bool)) for v in a)]}")
            try:
                return func(*args, **(kwargs or {}))
            finally:
                torch.cuda.nvtx.range_pop()

    with torch.autograd.profiler.emit_nvtx(record_shapes=False), Shapes():
        torch.cuda.nvtx.range_push("COMBO_MARKED_STEP")
        fwd_bwd()
        torch.cuda.synchronize()
        torch.cuda.nvtx.range_pop()


def parse(d):
    api = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
    mk = glob.glob(os.path.join(d, "**", "*marker_api_trace.csv"), recursive=True)[0]
    a = list(csv.DictReader(open(api)))
    m = list(csv.DictReader(open(mk)))
    s_key = [k for k in a[0] if k.lower().startswith("start")][0]
    e_key = [k for k in a[0] if k.lower().startswith("end")][0]
    ranges = [(int(r[s_key]), int(r[e_key]), r["Function"]) for r in m]
    step = [r for r in ranges if r[2] == "COMBO_MARKED_STEP"]
    if not step:
        print("marked step not found; marker functions seen:", collections.Counter(r[2] for r in ranges).most_common(5))
        return
    lo, hi = step[0][0], step[0][1]
    inner = sorted((r for r in ranges if lo <= r[0] and r[1] <= hi and r[2] != "COMBO_MARKED_STEP"), key=lambda r: r[0])
    agg = collections.Counter()
    for r in a:
        f = r["Function"]
        if not ("emset" in f or f == "hipMemcpyAsync"):
            continue
        t = int(r[s_key])
        if not (lo <= t <= hi):
            continue
        around = [x for x in inner if x[0] <= t <= x[1]]
        around.sort(key=lambda x: x[1] - x[0])
        label = " <- ".join((x[2] if x[2].startswith("SHAPES") else x[2].split(",")[0])[:150] for x in around[:3]) or "(no range)"
        agg[(f, label)] += 1
    for (f, where), n in agg.most_common():
        print(f"{n:5d} x {f:16s} {where}")


def kernels(d, pattern):
    """which ops launch the kernels whose name matches `pattern`?  Needs --kernel-trace as well:
    rocprofv3 --hip-trace --kernel-trace --marker-trace --output-format csv -d DIR -o t -- python3 tools/memset_sites.py run CFG
    python3 tools/memset_sites.py kernels DIR direct_copy"""
    import re
    api = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0])))
    mk = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*marker_api_trace.csv"), recursive=True)[0])))
    kt = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0])))
    s_key = [k for k in api[0] if k.lower().startswith("start")][0]
    e_key = [k for k in api[0] if k.lower().startswith("end")][0]
    corr = [k for k in api[0] if "orrelation" in k][0]
    kcorr = [k for k in kt[0] if "orrelation" in k][0]
    kname = [k for k in kt[0] if k.lower() in ("kernel_name", "name")][0]
    launch_time = {r[corr]: int(r[s_key]) for r in api}
    ranges = [(int(r[s_key]), int(r[e_key]), r["Function"]) for r in mk]
    step = [r for r in ranges if r[2] == "COMBO_MARKED_STEP"][0]
    inner = sorted((r for r in ranges if step[0] <= r[0] and r[1] <= step[1] and r[2] != "COMBO_MARKED_STEP"), key=lambda r: r[0])
    agg = collections.Counter()
    rx = re.compile(pattern)
    for r in kt:
        if not rx.search(r[kname]):
            continue
        t = launch_time.get(r[kcorr])
        if t is None or not (step[0] <= t <= step[1]):
            continue
        around = sorted((x for x in inner if x[0] <= t <= x[1]), key=lambda x: x[1] - x[0])
        agg[" <- ".join((x[2] if x[2].startswith("SHAPES") else x[2].split(",")[0])[:110] for x in around[:3]) or "(no range)"] += 1
    for where, n in agg.most_common(int(os.environ.get("TOP", "25"))):
        print(f"{n:5d} x {where}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    elif sys.argv[1] == "kernels":
        kernels(sys.argv[2], sys.argv[3])
    else:
        parse(sys.argv[2])
