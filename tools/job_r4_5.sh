mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_conv3x3_gpu.py -q -x -m gpu > gpurun_out/t6.log 2>&1; echo rc=$? >> gpurun_out/t6.log
tail -3 gpurun_out/t6.log
timeout 300 python tools/bench_nt2.py --no-lib > gpurun_out/nt3_bench.txt 2>&1
COMBO_DX_KERNEL=2 timeout 300 python tools/bench_nt2.py --no-lib > gpurun_out/nt2_bench_all.txt 2>&1
rm -f gpurun_out/nt3_abl.txt
for d in 1 2 4 8 16 32 63; do echo "== COMBO_NT3_DBG=$d" >> gpurun_out/nt3_abl.txt; COMBO_NT3_DBG=$d timeout 200 python tools/bench_nt2.py --shapes small --no-lib >> gpurun_out/nt3_abl.txt 2>&1; done
cat gpurun_out/nt3_bench.txt
