#!/bin/bash
# round 5, call 1: the new boundary / parity tests + a baseline bench line
mkdir -p gpurun_out
python -m pytest tests/test_msda_module.py tests/test_dp_gpu.py tests/test_backbone_x3_gpu.py -x -q -m gpu > gpurun_out/r5_1_tests_a.log 2>&1
echo "tests_a rc $?" >> gpurun_out/r5_1_tests_a.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "mask_logits_of_the_whole_model or training_forward_matches or eval_output" > gpurun_out/r5_1_tests_b.log 2>&1
echo "tests_b rc $?" >> gpurun_out/r5_1_tests_b.log
python -m pytest tests/test_head_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r5_1_tests_c.log 2>&1
echo "tests_c rc $?" >> gpurun_out/r5_1_tests_c.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_1_bench.json 2> gpurun_out/r5_1_bench.err
echo "bench rc $?" >> gpurun_out/r5_1_bench.err
tail -3 gpurun_out/r5_1_tests_a.log gpurun_out/r5_1_tests_b.log gpurun_out/r5_1_tests_c.log
