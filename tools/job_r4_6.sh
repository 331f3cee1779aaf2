mkdir -p gpurun_out
rm -f gpurun_out/nt3_abl.txt
for d in 0 1 2 4 8 16 32 41 63; do echo "== COMBO_NT3_DBG=$d" >> gpurun_out/nt3_abl.txt; COMBO_NT3_DBG=$d timeout 200 python tools/bench_nt3.py --shapes small --no-lib >> gpurun_out/nt3_abl.txt 2>&1; done
grep -v amdgpu gpurun_out/nt3_abl.txt | sed 's/floors.*//' | cut -c1-100
