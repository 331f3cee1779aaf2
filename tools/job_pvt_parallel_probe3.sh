#!/bin/bash
# round 6: the two-stream PVT capture with the runtime's graph packet capture left ON (the package normally switches it off)
mkdir -p gpurun_out/r6
P="combo_avs_amd.backbone_pvt.PyramidVisionTransformerV2.concurrent_safe=1"
B="--config pvt_ms3_t10 --no-cpu-baseline --no-other-workloads --no-exclusive --steps 5 --warmup 2"
out=gpurun_out/r6/pvt_par3.txt; : > $out
echo "== two streams, DEBUG_CLR_GRAPH_PACKET_CAPTURE left at the runtime's default (COMBO_GRAPH_MEMSET_GUARD=0 COMBO_ALLOW_PACKET_CAPTURE=1), tuning off" >> $out
COMBO_GRAPH_MEMSET_GUARD=0 COMBO_ALLOW_PACKET_CAPTURE=1 COMBO_GEMM_TUNING=0 COMBO_BENCH_TRACE=1 timeout 260 python tools/run_with_dump.py 180 $P -- $B > gpurun_out/r6/pq.out 2> gpurun_out/r6/pq.err
tail -1 gpurun_out/r6/pq.out | cut -c1-170 >> $out; grep "Timeout\|Error\|error" gpurun_out/r6/pq.err | head -3 | cut -c1-200 >> $out
cat $out
