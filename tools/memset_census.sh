#!/bin/bash
# How many hipMemset* calls does a training run make?  Inside a stream capture each becomes a memset NODE of the hipGraph, and
# those do not replay reliably on this stack (tools/graph_reduce_repro.py), so the count over a short graphed bench run should
# be zero (or all of them outside the captured region).   usage: tools/memset_census.sh [bench config]   (on the GPU box)
cfg=${1:-r50_s4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/memset_$cfg
rocprofv3 --hip-trace --stats -d /tmp/memset_$cfg -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 2 --warmup 2 --no-cpu-baseline > /tmp/memset_$cfg.log 2>&1
f=$(find /tmp/memset_$cfg -name '*hip_api_stats.csv' -o -name '*hip_stats.csv' | head -1)
[ -z "$f" ] && { echo "no HIP API stats file; wrote:"; find /tmp/memset_$cfg -type f | head; tail -3 /tmp/memset_$cfg.log; exit 1; }
echo "== $cfg: hipMemset* calls (name, calls)"
grep -i memset "$f" | cut -d, -f1,2 || echo "none"
grep -i -E "hipGraphLaunch|hipMemcpyAsync|hipLaunchKernel|hipModuleLaunchKernel|hipExtModuleLaunchKernel" "$f" | cut -d, -f1,2
