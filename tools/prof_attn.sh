#!/bin/bash
# per-kernel durations of the decoder attention kernels (rocprofv3 kernel trace; output to files, never through a pipe)
cd /tmp && export TMPDIR=/tmp
for lk in ${@:-784}; do
  rm -rf /tmp/pa
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/pa -o pa --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_attn.py $lk > /tmp/pa.log 2>&1 < /dev/null
  f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
  echo "Lk=$lk (dbg=${COMBO_ATTN_DBG:-0})"; grep "attn_" $f | awk -F, '{printf "  %-60s calls %s avg_us %.1f min %.1f max %.1f\n", substr($1,1,60), $2, $4/1e3, $6/1e3, $7/1e3}'
done
