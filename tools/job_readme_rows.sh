#!/bin/bash
# the optional modes of README's table, one bench line each (not the quoted metric)
mkdir -p gpurun_out; rm -f gpurun_out/readme_rows.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads"
run() { timeout 500 python bench.py $B "$@" 2> gpurun_out/rr.err | grep '^{"metric"' > gpurun_out/rr.json; python - "$*" <<'PY' >> gpurun_out/readme_rows.txt
import json,sys
try:
    d=json.loads(open("gpurun_out/rr.json").read().strip().splitlines()[-1]); print(sys.argv[1], "|", d["value"], d["ms_per_step"], d["unit"])
except Exception as e: print(sys.argv[1], "| FAILED", e)
PY
}
run
run --head-dtype bf16
run --head-dtype x3
run --dtype bf16
run --mode infer
run --config pvt_s4
cat gpurun_out/readme_rows.txt
