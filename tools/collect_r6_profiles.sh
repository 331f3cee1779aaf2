#!/bin/bash
# copies the outputs of tools/job_r6_final.sh (gpurun_out/) into profiles/ under the round's names
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import json
line=[l for l in open("gpurun_out/final_bench.json").read().strip().splitlines() if l.startswith('{"metric"')][-1]
json.dump(json.loads(line), open("profiles/r06_bench_bs8_default_run.json","w"), indent=1)
line=[l for l in open("gpurun_out/prof_bench_line.json").read().strip().splitlines() if l.startswith('{"metric"')][-1]
json.dump(json.loads(line), open("profiles/r06_bench_bs8_under_rocprofv3.json","w"), indent=1)
PY
cp gpurun_out/steady_graph.csv profiles/r06_bench_bs8_hipgraph_steady_state_kernel_summary.csv
cp gpurun_out/kstats.csv profiles/r06_bench_bs8_rocprofv3_kernel_stats.csv
cp gpurun_out/r06_pmc.json profiles/r06_pmc.json
cp gpurun_out/pmc_bench.txt profiles/r06_pmc_bench_counters.txt
cp gpurun_out/steady_pvt_ms3_t10.csv profiles/r06_pvt_ms3_t10_steady_state_kernel_summary.csv
cp gpurun_out/steady_pvt_avss_512.csv profiles/r06_pvt_avss_512_steady_state_kernel_summary.csv
if [ -f gpurun_out/slots_single.err ]; then
  (echo "# python bench.py --single-stream --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --dump-slots 2 (round 6, final sources): every gemm_nt3 launch"; echo "# of the timed hipGraph replays (ONE stream: the launches have the chip to themselves), grouped by (useful flops, algorithmic bytes); device-side duration per launch."; python tools/slots_by_shape.py gpurun_out/slots_single.err 2) > profiles/r06_nt3_in_step_by_shape.txt
fi
grep -E "passed|failed" gpurun_out/final_tests.log | tail -1
ls -la profiles/r06_*
