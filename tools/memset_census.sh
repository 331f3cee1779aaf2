#!/bin/bash
# How many hipMemset* calls does a training run make?  Inside a stream capture each becomes a memset NODE of the hipGraph, and
# those do not replay reliably on this stack (tools/graph_reduce_repro.py), so the count over a short graphed bench run should
# be zero (or all of them outside the captured region).   usage: tools/memset_census.sh [bench config]   (on the GPU box)
cfg=${1:-r50_s4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/memset_$cfg
rocprofv3 --hip-trace --stats --output-format csv -d /tmp/memset_$cfg -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 2 --warmup 2 --no-cpu-baseline > /tmp/memset_$cfg.log 2>&1
t=$(find /tmp/memset_$cfg -name '*hip_api_trace.csv' | head -1)
[ -z "$t" ] && { echo "no HIP API trace; wrote:"; find /tmp/memset_$cfg -type f | head; tail -3 /tmp/memset_$cfg.log; exit 1; }
python3 - "$t" "$cfg" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name = "Function" if "Function" in rows[0] else [k for k in rows[0] if "unction" in k or "Name" in k][0]
start = [k for k in rows[0] if k.lower().startswith("start")][0]
rows.sort(key=lambda r: int(r[start]))
inside, depth = collections.Counter(), 0
total = collections.Counter()
for r in rows:
    f = r[name]
    if f == "hipStreamBeginCapture":
        depth += 1
    elif f == "hipStreamEndCapture":
        depth -= 1
    elif "emset" in f or f.startswith("hipMemcpy"):
        total[f] += 1
        if depth > 0:
            inside[f] += 1
print(f"== {sys.argv[2]}: memset / memcpy API calls of the whole run: {dict(total)}")
print(f"== {sys.argv[2]}: of those INSIDE a stream capture (= nodes of the replayed hipGraph): {dict(inside) or 'none'}")
PY
