#!/bin/bash
# round 6, closing pass at the final csrc digest: PMC traffic (bench.py reads profiles/r06_pmc.json of the same digest), kernel-level suites, the default bench run
mkdir -p gpurun_out profiles
export COMBO_COMMIT=$(cat .combo_commit 2>/dev/null || echo unknown)
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
cp gpurun_out/r06_pmc.json profiles/r06_pmc.json 2>/dev/null
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_conv3x3_gpu.py tests/test_f16x3_gpu.py tests/test_backbone_x3_gpu.py tests/test_kernels_gpu.py -q -m gpu > gpurun_out/close_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/close_tests.log
tail -2 gpurun_out/close_tests.log
timeout 900 python bench.py > gpurun_out/close_bench.json 2> gpurun_out/close_bench.err
tail -c 200 gpurun_out/close_bench.json
