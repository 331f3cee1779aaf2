#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_msda_gpu.py -x -q -m gpu > gpurun_out/r5_3_tests_msda.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_3_tests_msda.log
bash tools/abl_msda_bwd.sh > gpurun_out/r5_3_msda_bwd_ablation.txt 2>&1
python tools/bench_msda.py --iters 100 > gpurun_out/r5_3_msda_bench.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_3_bench.json 2> gpurun_out/r5_3_bench.err
echo "bench rc $?" >> gpurun_out/r5_3_bench.err
tail -n 3 gpurun_out/r5_3_tests_msda.log; cat gpurun_out/r5_3_msda_bwd_ablation.txt
