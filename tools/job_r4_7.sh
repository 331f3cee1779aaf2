mkdir -p gpurun_out
rm -f gpurun_out/nt3_round.txt
for t in 1 2 3; do echo "== tile $t" >> gpurun_out/nt3_round.txt; timeout 200 python tools/bench_nt2.py --shapes round --no-lib --tile $t >> gpurun_out/nt3_round.txt 2>&1; done
for d in 1 2 8 16 32 41 63; do echo "== tile 1 COMBO_NT3_DBG=$d" >> gpurun_out/nt3_round.txt; COMBO_NT3_DBG=$d timeout 200 python tools/bench_nt2.py --shapes round --no-lib --tile 1 >> gpurun_out/nt3_round.txt 2>&1; done
echo "== nt2" >> gpurun_out/nt3_round.txt; COMBO_DX_KERNEL=2 timeout 200 python tools/bench_nt2.py --shapes round --no-lib >> gpurun_out/nt3_round.txt 2>&1
grep -v amdgpu gpurun_out/nt3_round.txt | sed 's/floors.*//' | cut -c1-100
