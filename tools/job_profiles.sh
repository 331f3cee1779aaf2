mkdir -p gpurun_out
# the commit the counters are stamped with: git here; on the GPU box (no .git) the file tools/grun.sh wrote before the snapshot
export COMBO_COMMIT=$(git rev-parse --short HEAD 2>/dev/null || cat .combo_commit 2>/dev/null || echo unknown)
bash tools/final_profile.sh --profile-only > gpurun_out/final_profile.log 2>&1
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
tail -3 gpurun_out/pmc_bench.log | cut -c1-400
bash tools/prof_config.sh pvt_ms3_t10 > gpurun_out/prof_ms3.log 2>&1
bash tools/prof_config.sh pvt_avss_512 3 2 > gpurun_out/prof_avss.log 2>&1
python tools/bench_r50_x3.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r50_x3.txt
python tools/bench_pointlogits.py 2>&1 | grep -v amdgpu.ids > gpurun_out/pointlogits.txt
python tools/bench_nt3.py --shapes all 2>&1 | grep -v amdgpu.ids > gpurun_out/nt3_bench.txt
python tools/bench_r50_convs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r50_convs.txt
ls -la gpurun_out/kstats.csv gpurun_out/steady_graph.csv gpurun_out/steady_pvt_ms3_t10.csv gpurun_out/steady_pvt_avss_512.csv gpurun_out/r05_pmc.json gpurun_out/prof_bench_line.json
