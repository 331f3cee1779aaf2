mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_conv3x3_gpu.py -q -x -m gpu > gpurun_out/t6.log 2>&1; echo rc=$? >> gpurun_out/t6.log
tail -3 gpurun_out/t6.log
rm -f gpurun_out/nt3_round.txt
for t in 1 2; do echo "== tile $t" >> gpurun_out/nt3_round.txt; timeout 200 python tools/bench_nt3.py --shapes round --no-lib --tile $t >> gpurun_out/nt3_round.txt 2>&1; done
for d in 1 2 8 32 41; do echo "== tile 1 COMBO_NT3_DBG=$d" >> gpurun_out/nt3_round.txt; COMBO_NT3_DBG=$d timeout 200 python tools/bench_nt3.py --shapes round --no-lib --tile 1 >> gpurun_out/nt3_round.txt 2>&1; done
grep -v amdgpu gpurun_out/nt3_round.txt | sed 's/floors.*//' | cut -c1-100
timeout 300 python tools/bench_nt3.py --no-lib > gpurun_out/nt3_bench.txt 2>&1
grep -v amdgpu gpurun_out/nt3_bench.txt | sed 's/floors.*//' | cut -c1-100
