#!/usr/bin/env python3
"""stderr of `python bench.py ... --dump-slots <kind>` -> the launches of that kind grouped by (useful flops, algorithmic bytes):
    python tools/slots_by_shape.py gpurun_out/slots_bench.err 2 > profiles/rNN_nt3_in_step_by_shape.txt"""
import collections
import re
import sys

path, kind = sys.argv[1], sys.argv[2]
g = collections.OrderedDict()
for line in open(path):
    m = re.match(r"\[slot (\d+)\] kind (\d+) work ([\d.e+-]+) bytes ([\d.e+-]+) avg_us ([\d.]+)", line)
    if m and m.group(2) == kind:
        w, b, us = float(m.group(3)), float(m.group(4)), float(m.group(5))
        k = (round(w / 1e9, 2), round(b / 1e6, 1))
        d = g.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += us
rows = sorted(g.items(), key=lambda kv: -kv[1][1])
n = sum(v[0] for _, v in rows)
tot = sum(v[1] for _, v in rows)
print(f"# {n} launches per step, {tot / 1e3:.2f} ms per step.")
for (gf, mb), (cnt, us) in rows:
    avg = us / cnt
    print(f"GF {gf:7.2f}  MB {mb:7.1f}  x {cnt:3d}  avg {avg:7.1f} us  total {us:7.0f} us  {gf / avg * 1e3 if avg else 0:7.1f} TF/s useful  {mb / avg * 1e3 if avg else 0:7.0f} GB/s")
