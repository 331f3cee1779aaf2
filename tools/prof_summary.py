#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV over the steady-state window only.

    python tools/prof_summary.py <kernel_trace.csv> <steps> [marker_kernel_substring] [calls_per_step]

MIOpen's find phase (first call of every conv config) runs dozens of candidate kernels, which pollutes
`--stats`.  This takes the last `steps` training steps (delimited by the marker kernel, default the MSDeformAttn
forward core which runs 6x per step) and prints per-kernel totals, plus the GPU-busy time per step."""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 else "msda_fwd_tap_d32"
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
assert len(marks) >= per_step * (steps + 1), (len(marks), per_step, steps)
# window: from the first marker of the (steps)-th step from the end back to ... we approximate a step boundary by the
# first marker launch of a step; use [first marker of step -steps, first marker of the trailing partial step)
first = marks[len(marks) - per_step * steps - per_step]
last = marks[len(marks) - per_step]
t0, t1 = rows[first][0], rows[last][0]
agg = defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in rows[first:last]:
    agg[n][0] += 1
    agg[n][1] += e - s
    busy += e - s
tot = sum(v[1] for v in agg.values())
print(f"# window: {steps} steps, wall {(t1 - t0) / steps / 1e6:.3f} ms/step, sum of kernel durations {busy / steps / 1e6:.3f} ms/step, "
      f"{sum(v[0] for v in agg.values()) / steps:.0f} kernel launches/step")
print("name,calls_per_step,total_ms_per_step,avg_us,percent")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"\"{n[:150]}\",{c / steps:.1f},{d / steps / 1e6:.3f},{d / c / 1e3:.1f},{100.0 * d / tot:.2f}")
