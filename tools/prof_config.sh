#!/bin/bash
# rocprofv3 kernel trace of `bench.py --config $1` (output to files, never through a pipe) -> steady-state per-kernel summary of
# the graph-replayed steps (tools/prof_graph_steps.py) in gpurun_out/steady_$1.csv
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/pc
rocprofv3 --kernel-trace --stats -d /tmp/pc -o c --output-format csv -- python3 bench.py --config $1 --steps ${2:-4} --warmup 2 --no-cpu-baseline > gpurun_out/prof_$1.log 2>&1 < /dev/null
t=$(find /tmp/pc -name 'c_kernel_trace.csv' | head -1)
python tools/prof_graph_steps.py "$t" ${3:-3} 0 > gpurun_out/steady_$1.csv
head -32 gpurun_out/steady_$1.csv | cut -c1-150
