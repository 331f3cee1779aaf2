#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_backbone_x3_gpu.py tests/test_conv3x3_gpu.py tests/test_kernels_gpu.py -q -m gpu > gpurun_out/r5_13_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_13_tests.log
python -m pytest tests/test_model_gpu.py -q -m gpu -k "one_train_step" >> gpurun_out/r5_13_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_13_tests.log
grep -E "passed|failed|FAILED|rc" gpurun_out/r5_13_tests.log | tail -12
