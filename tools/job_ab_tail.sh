#!/bin/bash
# tests touched by the round-5 tail items + a bench line on the same box with and without them
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_head_gpu.py tests/test_graph_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/tail_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tail_tests.log
B="--no-cpu-baseline --no-other-workloads --steps 30 --warmup 5"
rm -f gpurun_out/tail_ab.txt
for v in "x.y=0" "combo_avs_amd.modeling.layers.MLP_FUSED_RELU_GRAD=0" "x.y=0" "combo_avs_amd.modeling.layers.MLP_FUSED_RELU_GRAD=0"; do
  if [ "$v" = "x.y=0" ]; then
    timeout 400 python bench.py $B > gpurun_out/tail_tmp.json 2> gpurun_out/tail_tmp.err
  else
    timeout 400 python tools/ab_const.py $v -- $B > gpurun_out/tail_tmp.json 2> gpurun_out/tail_tmp.err
  fi
  python - "$v" <<'PY' >> gpurun_out/tail_ab.txt
import json,sys
try:
    d=json.loads(open("gpurun_out/tail_tmp.json").read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d.get("ms_per_step_median"), d.get("launch"))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("gpurun_out/tail_tmp.err").read()[-1500:])
PY
done
tail -4 gpurun_out/tail_tests.log; cat gpurun_out/tail_ab.txt
