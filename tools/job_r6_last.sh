#!/bin/bash
# round 6, last pass at the final sources: the whole GPU suite, the default bench run (JSON line -> profiles/), README's optional rows
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/last_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/last_tests.log
tail -3 gpurun_out/last_tests.log
timeout 900 python bench.py > gpurun_out/last_bench.json 2> gpurun_out/last_bench.err
tail -c 300 gpurun_out/last_bench.json
B="--steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-exclusive"
rm -f gpurun_out/readme_rows.txt
run() { timeout 500 python bench.py $B "$@" 2> gpurun_out/rr.err | grep '^{"metric"' > gpurun_out/rr.json; python - "$*" <<'PY' >> gpurun_out/readme_rows.txt
import json,sys
try:
    d=json.loads(open("gpurun_out/rr.json").read().strip().splitlines()[-1]); print(sys.argv[1], "|", d["value"], d["ms_per_step"], d["unit"])
except Exception as e: print(sys.argv[1], "| FAILED", e)
PY
}
run --head-dtype f16x3
run --head-dtype fp32
run --head-dtype bf16
run --head-dtype x3
run --dtype bf16
run --mode infer
run --config pvt_s4
cat gpurun_out/readme_rows.txt
