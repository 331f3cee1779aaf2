#!/usr/bin/env python3
"""Duration histogram of the launches of one kernel (name substring) inside the graph-replayed steps of a rocprofv3 kernel trace:
    python tools/prof_hist.py <kernel_trace.csv> <substring> [steps=5]
prints launches per step and total time per duration bucket (a proxy for the tensor sizes behind an element-wise kernel)."""
import csv, sys
from collections import defaultdict
path, sub = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "msda_fwd_tap_d32" in r[2]] or [i for i, r in enumerate(rows) if "msda_fwd_generic" in r[2]]
per = 6
first, last = marks[len(marks) - per * (steps + 1)], marks[len(marks) - per]
edges = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 1e9]
agg = defaultdict(lambda: [0, 0.0])
for s, e, n in rows[first:last]:
    if sub in n:
        us = (e - s) / 1e3
        b = next(x for x in edges if us <= x)
        agg[b][0] += 1
        agg[b][1] += us
for b in sorted(agg):
    print(f"<= {b:>6} us: {agg[b][0] / steps:7.1f} launches/step  {agg[b][1] / steps / 1e3:7.3f} ms/step")
