#!/bin/bash
# rocprofv3 kernel stats of `bench.py --config $1` (output to files, never through a pipe); prints the top kernels
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/pc
rocprofv3 --kernel-trace --stats -d /tmp/pc -o c --output-format csv -- python3 bench.py --config $1 --steps ${2:-3} --warmup 2 --no-cpu-baseline > gpurun_out/prof_$1.log 2>&1 < /dev/null
f=$(find /tmp/pc -name 'c_kernel_stats.csv' | head -1); cp "$f" gpurun_out/kstats_$1.csv
head -25 gpurun_out/kstats_$1.csv | awk -F'","' '{printf "%-90s calls %s total_ms %.1f avg_us %.1f pct %s\n", substr($1,2,90), $2, $3/1e6, $4/1e3, $5}'
