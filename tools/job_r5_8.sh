#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_sra_gpu.py -q -m gpu > gpurun_out/r5_8_tests_sra.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_8_tests_sra.log
python tools/dbg_sra2.py > gpurun_out/r5_8_dbg.txt 2>&1
python tools/bench_sra.py > gpurun_out/r5_8_sra_bench.txt 2>&1
python -m pytest tests/test_criterion_padded_gpu.py tests/test_gemm_gpu.py -q -m gpu -x > gpurun_out/r5_8_tests_b.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_8_tests_b.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "avss" > gpurun_out/r5_8_tests_avss.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_8_tests_avss.log
tail -n 4 gpurun_out/r5_8_tests_sra.log; tail -n 2 gpurun_out/r5_8_dbg.txt; grep "^\[" gpurun_out/r5_8_sra_bench.txt; tail -n 5 gpurun_out/r5_8_tests_b.log; tail -n 5 gpurun_out/r5_8_tests_avss.log
