#!/bin/bash
# quick confirmation at the working tree: the f16x3 tests, the head golden tests, a short bench A/B of the head forward modes
mkdir -p gpurun_out/quick
o=gpurun_out/quick
COMBO_TEST_VERBOSE=1 timeout 900 python -m pytest tests/test_f16x3_gpu.py tests/test_head_gpu.py -q -m gpu -x > $o/tests.log 2>&1
echo "rc=$?" >> $o/tests.log
grep -E "passed|failed|^FAILED|^E  +Assert|rc=" $o/tests.log | cut -c1-300 | tail -8
B="--steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --no-exclusive"
for m in f16x3 fp32 f16x3 fp32; do
  timeout 600 python bench.py $B --head-dtype $m > $o/bench_$m.json 2> $o/bench_$m.err
  python - $o/bench_$m.json $m <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["dtype"])
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
