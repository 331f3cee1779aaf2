mkdir -p gpurun_out/r6

# PVT backbone has no concurrent_safe attribute: force via class attribute on the PVT class
P="combo_avs_amd.backbone_pvt.PyramidVisionTransformerV2.concurrent_safe=1"
B="--config pvt_ms3_t10 --no-cpu-baseline --no-other-workloads --no-exclusive --steps 5 --warmup 2"
echo "== graph, tuning off" > gpurun_out/r6/pvt_par.txt
COMBO_GEMM_TUNING=0 COMBO_BENCH_TRACE=1 timeout 250 python tools/run_with_dump.py 170 $P -- $B > gpurun_out/r6/pp1.out 2> gpurun_out/r6/pp1.err; tail -1 gpurun_out/r6/pp1.out | cut -c1-200 >> gpurun_out/r6/pvt_par.txt; grep "bench rank\|Timeout" gpurun_out/r6/pp1.err | cut -c1-120 >> gpurun_out/r6/pvt_par.txt
echo "== eager, tuning on" >> gpurun_out/r6/pvt_par.txt
COMBO_BENCH_TRACE=1 timeout 300 python tools/run_with_dump.py 220 $P -- $B --no-graph > gpurun_out/r6/pp2.out 2> gpurun_out/r6/pp2.err; tail -1 gpurun_out/r6/pp2.out | cut -c1-200 >> gpurun_out/r6/pvt_par.txt; grep "bench rank\|Timeout" gpurun_out/r6/pp2.err | cut -c1-120 >> gpurun_out/r6/pvt_par.txt
echo "== single stream graph, tuning off (reference)" >> gpurun_out/r6/pvt_par.txt
COMBO_GEMM_TUNING=0 timeout 250 python bench.py $B > gpurun_out/r6/pp3.out 2> gpurun_out/r6/pp3.err; tail -1 gpurun_out/r6/pp3.out | cut -c1-200 >> gpurun_out/r6/pvt_par.txt
cat gpurun_out/r6/pvt_par.txt
