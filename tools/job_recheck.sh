#!/bin/bash
# re-run of the two tests the final job failed (they pinned the exact path's launch names) + the suites around the forward mode, at the final sources
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv3x3_gpu.py tests/test_kernels_gpu.py tests/test_f16x3_gpu.py tests/test_head_gpu.py tests/test_gemm_gpu.py -q -m gpu > gpurun_out/recheck_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/recheck_tests.log
tail -3 gpurun_out/recheck_tests.log
