#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py tests/test_head_gpu.py tests/test_backbone_x3_gpu.py tests/test_conv3x3_gpu.py -q -m gpu > gpurun_out/r5_15_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_15_tests.log
: > gpurun_out/r5_15_ab.txt
for rep in 1 2; do
  for mode in 1 0; do
    python tools/ab_const.py call:combo_gemm_tn_tile256=$mode -- --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j = json.loads(l); k = j['other_kernels'].get('gemm_tn_grouped_kernel', j['roofline'])
        print('tile256=$mode rep $rep', j['value'], 'frames/s', j['ms_per_step'], 'ms per step; grouped dW', k['ms_per_step'], 'ms', k['frac'])
" >> gpurun_out/r5_15_ab.txt
  done
done
grep -E "passed|failed|FAILED|rc" gpurun_out/r5_15_tests.log | tail -8; cat gpurun_out/r5_15_ab.txt
