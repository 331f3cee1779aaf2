#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_11_tests_all.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_11_tests_all.log
( time python bench.py ) > gpurun_out/r5_11_bench_default.json 2> gpurun_out/r5_11_bench_default.err
echo "bench rc $?" >> gpurun_out/r5_11_bench_default.err
tail -n 6 gpurun_out/r5_11_tests_all.log; tail -n 6 gpurun_out/r5_11_bench_default.err; head -c 300 gpurun_out/r5_11_bench_default.json
