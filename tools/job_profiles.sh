mkdir -p gpurun_out
export COMBO_COMMIT=3e99ec4
bash tools/final_profile.sh --profile-only > gpurun_out/final_profile.log 2>&1
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
tail -3 gpurun_out/pmc_bench.log | cut -c1-400
bash tools/prof_config.sh pvt_ms3_t10 > gpurun_out/prof_ms3.log 2>&1
bash tools/prof_config.sh pvt_avss_512 3 2 > gpurun_out/prof_avss.log 2>&1
python tools/bench_r50_x3.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r50_x3.txt
python tools/bench_pointlogits.py 2>&1 | grep -v amdgpu.ids > gpurun_out/pointlogits.txt
python tools/bench_nt3.py --shapes all 2>&1 | grep -v amdgpu.ids > gpurun_out/nt3_bench.txt
python tools/bench_r50_convs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r50_convs.txt
ls -la gpurun_out/kstats.csv gpurun_out/steady_graph.csv gpurun_out/steady_pvt_ms3_t10.csv gpurun_out/steady_pvt_avss_512.csv gpurun_out/r04_pmc.json gpurun_out/prof_bench_line.json
