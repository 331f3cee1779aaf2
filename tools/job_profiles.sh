mkdir -p gpurun_out
export COMBO_COMMIT=ee386db
bash tools/final_profile.sh --profile-only > gpurun_out/final_profile.log 2>&1
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
tail -3 gpurun_out/pmc_bench.log | cut -c1-600
bash tools/prof_config.sh pvt_ms3_t10 > gpurun_out/prof_ms3.log 2>&1
bash tools/prof_config.sh pvt_avss_512 3 2 > gpurun_out/prof_avss.log 2>&1
ls -la gpurun_out/kstats.csv gpurun_out/steady_graph.csv gpurun_out/steady_pvt_ms3_t10.csv gpurun_out/steady_pvt_avss_512.csv gpurun_out/r04_pmc.json gpurun_out/prof_bench_line.json
