#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gemm_gpu.py tests/test_backbone_x3_gpu.py tests/test_conv3x3_gpu.py -q -m gpu -x > gpurun_out/r5_10_tests_a.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_10_tests_a.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "avss" > gpurun_out/r5_10_tests_avss.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_10_tests_avss.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-slots 2 > gpurun_out/r5_10_bench.json 2> gpurun_out/r5_10_bench.err
echo "bench rc $?" >> gpurun_out/r5_10_bench.err
python bench.py --config pvt_ms3_t10 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r5_10_bench_ms3.json 2> gpurun_out/r5_10_bench_ms3.err
COMBO_BENCH_TRACE=1 python bench.py --config pvt_avss_512 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r5_10_bench_avss.json 2> gpurun_out/r5_10_bench_avss.err
echo "bench rc $?" >> gpurun_out/r5_10_bench_avss.err
tail -n 3 gpurun_out/r5_10_tests_a.log; tail -n 4 gpurun_out/r5_10_tests_avss.log; head -c 250 gpurun_out/r5_10_bench.json; echo; head -c 250 gpurun_out/r5_10_bench_ms3.json; echo; head -c 250 gpurun_out/r5_10_bench_avss.json; echo; grep -v "^\[bench rank\|^\[slot" gpurun_out/r5_10_bench_avss.err | tail -3
