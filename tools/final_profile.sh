#!/bin/bash
# Round-end measurement: the default bench line, then the same command under rocprofv3 (kernel trace + stats) and the
# steady-state (graph replay) summary.  Everything lands in gpurun_out/; copy what should be judged into profiles/.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$1" != "--profile-only" ]; then
  python bench.py 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json
  cut -c1-600 gpurun_out/bench_default.json
fi
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats -d /tmp/prof -o b --output-format csv -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-workloads > gpurun_out/prof_bench.log 2>&1
grep '^{"metric"' gpurun_out/prof_bench.log > gpurun_out/prof_bench_line.json
f=$(find /tmp/prof -name 'b_kernel_stats.csv' | head -1); cp "$f" gpurun_out/kstats.csv
t=$(find /tmp/prof -name 'b_kernel_trace.csv' | head -1)
python tools/prof_graph_steps.py "$t" 5 0 > gpurun_out/steady_graph.csv
head -60 gpurun_out/steady_graph.csv | cut -c1-170
if [ -n "$COMBO_PROF_HIST" ]; then python tools/prof_hist.py "$t" "$COMBO_PROF_HIST" 5 > gpurun_out/prof_hist.txt; fi
