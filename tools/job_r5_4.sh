#!/bin/bash
mkdir -p gpurun_out
tools/ubench/lds_atomic4 > gpurun_out/r5_4_lds_atomic4.txt 2>&1
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "avss" > gpurun_out/r5_4_tests_avss.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_4_tests_avss.log
COMBO_BENCH_TRACE=1 python bench.py --config pvt_avss_512 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r5_4_bench_avss.json 2> gpurun_out/r5_4_bench_avss.err
echo "bench rc $?" >> gpurun_out/r5_4_bench_avss.err
cat gpurun_out/r5_4_lds_atomic4.txt; tail -n 5 gpurun_out/r5_4_tests_avss.log; tail -n 12 gpurun_out/r5_4_bench_avss.err; head -c 400 gpurun_out/r5_4_bench_avss.json
