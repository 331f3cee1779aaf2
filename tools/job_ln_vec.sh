#!/bin/bash
# LayerNorm kernels (access width, parameter-gradient kernel): tests, then the three workloads
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_graph_gpu.py -x -q -k "prenorm or bias_ln or layernorm or pvt or graph or ln" > gpurun_out/ln_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/ln_tests.log
tail -3 gpurun_out/ln_tests.log
for c in r50_s4 pvt_ms3_t10 pvt_avss_512; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2> gpurun_out/ln_$c.err | grep '^{"metric"' > gpurun_out/ln_$c.json
  python - $c <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/ln_{sys.argv[1]}.json").read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
