#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu > gpurun_out/r5_12_tests_all.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_12_tests_all.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_12_bench.json 2> gpurun_out/r5_12_bench.err
tail -n 8 gpurun_out/r5_12_tests_all.log; head -c 300 gpurun_out/r5_12_bench.json
