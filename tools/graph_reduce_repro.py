#!/usr/bin/env python3
"""Isolated check of a hazard found in the captured training step (round 3): ATen reductions that split ONE output over several
workgroups (Reduce.cuh's global reduce: a staging buffer + per-output semaphores that a memset zeroes before the launch) come
back with stale memory now and then when they are replayed from a hipGraph and eager launches + device synchronisations sit
between the replays.  This script captures a handful of such reductions (the shapes of the PVTv2 bias gradients and of the
level-embedding gradient), replays the graph N times with an eager kernel between two synchronisations before every replay, and
counts the replays whose result differs from the eager result; then does the same for the replacement (`ops.colsum`).

usage: python tools/graph_reduce_repro.py [replays]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("SET_IN_PROCESS") == "1":  # what combo_avs_amd/__init__.py does: before the first HIP call of the process
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
import torch

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
torch.manual_seed(0)
shapes = [((15680, 1280), torch.bfloat16, (0,)), ((3920, 2048), torch.bfloat16, (0,)), ((80, 320, 14, 14), torch.bfloat16, (0, 2, 3)),
          ((80, 64, 56, 56), torch.bfloat16, (0, 2, 3)), ((40, 1029, 256), torch.float32, (0, 1)), ((15680, 320), torch.bfloat16, (0,)),
          ((8, 5376, 256), torch.float32, (0, 1))]
xs = [torch.randn(s, device=dev).to(dt) for s, dt, _ in shapes]


between = os.environ.get("BETWEEN", "kernel")  # "kernel": sync, one eager kernel, sync before every replay; "none"


def run(label, fn):
    want = [fn(x, d).float().clone() for x, (_, _, d) in zip(xs, shapes)]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            outs = [fn(x, d) for x, (_, _, d) in zip(xs, shapes)]
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        outs = []
        for rep in range(8):  # several launches of each, as a backward pass has
            outs.append([fn(x, d) for x, (_, _, d) in zip(xs, shapes)])
    junk = torch.ones(4096, device=dev)
    bad = 0
    worst = 0.0
    per_shape = [0] * len(shapes)
    for it in range(n_rep):
        if between == "kernel":
            torch.cuda.synchronize()
            junk[:16].zero_()
            torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        wrong = False
        for row in outs:
            for i, (o, w) in enumerate(zip(row, want)):
                err = float((o.float() - w).abs().max())
                if not (err <= 1e-2 * float(w.abs().max())):
                    wrong = True
                    per_shape[i] += 1
                    worst = max(worst, err) if err == err else float("inf")
        bad += wrong
    print(f"{label} [{between} between replays]: {bad} of {n_rep} replays returned a wrong reduction (worst abs error {worst:.3g}; "
          f"wrong results per shape {per_shape})", flush=True)
    return bad


bad_aten = run("ATen sum inside the graph", lambda x, d: x.sum(d))
try:
    import combo_avs_amd  # noqa: F401
    from combo_avs_amd.ops.colsum import sum_leading
    bad_own = run("ops.colsum inside the graph", lambda x, d: sum_leading(x, d))
    if os.environ.get("EXPECT_OWN_CLEAN") == "1" and bad_own:
        sys.exit(1)
except ImportError as e:
    print("ops.colsum not available:", e)
