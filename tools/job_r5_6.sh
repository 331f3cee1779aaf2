#!/bin/bash
mkdir -p gpurun_out
python tools/bench_sra.py > gpurun_out/r5_6_sra_bench.txt 2>&1
cat gpurun_out/r5_6_sra_bench.txt
