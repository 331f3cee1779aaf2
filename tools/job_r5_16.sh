#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_msda_gpu.py tests/test_msda_module.py tests/test_kernels_gpu.py tests/test_backbone_x3_gpu.py -q -m gpu > gpurun_out/r5_16_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_16_tests.log
python tools/bench_msda.py --iters 100 2>&1 | grep -v amdgpu > gpurun_out/r5_16_msda.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_16_bench.json 2> gpurun_out/r5_16_bench.err
grep -E "passed|failed|FAILED|rc" gpurun_out/r5_16_tests.log | tail -5; grep "windowed\|tap" gpurun_out/r5_16_msda.txt; python3 -c "
import json
j=json.loads([l for l in open('gpurun_out/r5_16_bench.json') if l.startswith('{\"metric\"')][-1])
print(j['value'], j['ms_per_step'], 'msda_bwd', j['other_kernels']['msda_bwd']['avg_launch_us'], j['other_kernels']['msda_bwd']['frac'])
"
