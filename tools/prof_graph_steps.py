#!/usr/bin/env python3
"""Summarise the hipGraph-replayed steps of a rocprofv3 --kernel-trace CSV of bench.py (graph mode).

    python tools/prof_graph_steps.py <kernel_trace.csv> [replayed_steps_to_use=5] [eager_tail_steps=2]

bench.py ends with `eager_tail_steps` eager re-runs (HIP-event timing of the MSDeformAttn launches); the steps before
are replays.  Steps are delimited by the MSDeformAttn forward kernel (6 launches per step)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "msda_fwd_tap_d32" in r[2]]
if not marks:  # pyramids that run the generic forward kernel (512 x 512 inputs)
    marks = [i for i, r in enumerate(rows) if "msda_fwd_generic" in r[2]]
per = 6
first, last = marks[len(marks) - per * tail - per * (steps + 1)], marks[len(marks) - per * tail - per * 1]
agg = defaultdict(lambda: [0, 0])
for s, e, n in rows[first:last]:
    agg[n][0] += 1
    agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"# hipGraph replays: {steps} steps, wall {(rows[last][0] - rows[first][0]) / steps / 1e6:.3f} ms/step, "
      f"sum of kernel durations {tot / steps / 1e6:.3f} ms/step, {sum(v[0] for v in agg.values()) / steps:.0f} kernel launches/step")
print("name,calls_per_step,total_ms_per_step,avg_us,percent")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"\"{n[:150]}\",{c / steps:.1f},{d / steps / 1e6:.3f},{d / c / 1e3:.1f},{100.0 * d / tot:.2f}")
