#!/bin/bash
# round 6: does the two-stream PVT capture hang because of the library GEMMs' workspace kernels (split-K / stream-K flags)?
mkdir -p gpurun_out/r6
P="combo_avs_amd.backbone_pvt.PyramidVisionTransformerV2.concurrent_safe=1"
B="--config pvt_ms3_t10 --no-cpu-baseline --no-other-workloads --no-exclusive --steps 5 --warmup 2"
out=gpurun_out/r6/pvt_par2.txt; : > $out
run() { # label, env...
  echo "== $1" >> $out; shift
  env "$@" COMBO_BENCH_TRACE=1 timeout 260 python tools/run_with_dump.py 180 $P -- $B > gpurun_out/r6/pq.out 2> gpurun_out/r6/pq.err
  tail -1 gpurun_out/r6/pq.out | cut -c1-170 >> $out; grep "Timeout\|Error\|error" gpurun_out/r6/pq.err | head -3 | cut -c1-200 >> $out
}
run "two streams, rocBLAS instead of hipBLASLt (TORCH_BLAS_PREFER_HIPBLASLT=0), tuning off" TORCH_BLAS_PREFER_HIPBLASLT=0 COMBO_GEMM_TUNING=0
run "two streams, no GEMM workspace (HIPBLASLT_WORKSPACE_SIZE=0 CUBLASLT_WORKSPACE_SIZE=0), tuning off" HIPBLASLT_WORKSPACE_SIZE=0 CUBLASLT_WORKSPACE_SIZE=0 COMBO_GEMM_TUNING=0
cat $out
