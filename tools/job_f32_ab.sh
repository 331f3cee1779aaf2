#!/bin/bash
# fp32 forward GEMM: correctness, micro-benchmark (with / without split-K), per-launch fixed cost probe, in-step per-shape times
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "f32 or linear_forward" > gpurun_out/f32_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/f32_tests.log
rm -f gpurun_out/f32_ab2.log
echo "=== default" >> gpurun_out/f32_ab2.log
timeout 300 python tools/bench_f32.py --no-lib >> gpurun_out/f32_ab2.log 2>&1
echo "=== nosplitk" >> gpurun_out/f32_ab2.log
timeout 300 python tools/bench_f32.py --no-lib --no-splitk >> gpurun_out/f32_ab2.log 2>&1
echo "=== nosplitk skinny" >> gpurun_out/f32_ab2.log
COMBO_F32_TILE=3 timeout 300 python tools/bench_f32.py --no-lib --no-splitk >> gpurun_out/f32_ab2.log 2>&1
echo "=== fixed" >> gpurun_out/f32_ab2.log
timeout 300 python tools/bench_f32.py --no-lib --shapes fixed >> gpurun_out/f32_ab2.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-other-workloads --dump-slots 1 > gpurun_out/f32_bench.json 2> gpurun_out/f32_slots.txt
tail -3 gpurun_out/f32_tests.log; tail -c 600 gpurun_out/f32_bench.json
