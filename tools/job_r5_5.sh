#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_sra_gpu.py -q -m gpu -s > gpurun_out/r5_5_tests_sra.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_5_tests_sra.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "avss" > gpurun_out/r5_5_tests_avss.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_5_tests_avss.log
COMBO_BENCH_TRACE=1 python bench.py --config pvt_avss_512 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r5_5_bench_avss.json 2> gpurun_out/r5_5_bench_avss.err
echo "bench rc $?" >> gpurun_out/r5_5_bench_avss.err
python bench.py --config pvt_ms3_t10 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r5_5_bench_ms3.json 2> gpurun_out/r5_5_bench_ms3.err
echo "bench rc $?" >> gpurun_out/r5_5_bench_ms3.err
grep -E "^\[sra|passed|failed|rc" gpurun_out/r5_5_tests_sra.log | tail -20; tail -n 4 gpurun_out/r5_5_tests_avss.log; grep -v "^\[bench rank" gpurun_out/r5_5_bench_avss.err | tail -4; head -c 300 gpurun_out/r5_5_bench_avss.json; echo; head -c 300 gpurun_out/r5_5_bench_ms3.json
