#!/bin/bash
# round 5 profile set: rocprofv3 kernel stats + steady-state summary of the default bench, PMC passes (HBM traffic), the two PVT
# workloads' steady-state summaries, micro-benchmarks cited in DESIGN.md.  Everything lands in gpurun_out/; copy into profiles/.
mkdir -p gpurun_out
export COMBO_COMMIT=$(git rev-parse --short HEAD 2>/dev/null || cat .combo_commit 2>/dev/null || echo unknown)
bash tools/final_profile.sh --profile-only > gpurun_out/final_profile.log 2>&1
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
tail -3 gpurun_out/pmc_bench.log | cut -c1-300
bash tools/prof_config.sh pvt_ms3_t10 > gpurun_out/prof_ms3.log 2>&1
bash tools/prof_config.sh pvt_avss_512 3 2 > gpurun_out/prof_avss.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-slots 2 > gpurun_out/slots_bench.json 2> gpurun_out/slots_bench.err
bash tools/abl_msda_bwd.sh > gpurun_out/msda_bwd_ablation.txt 2>&1
python tools/bench_msda.py --iters 100 >> gpurun_out/msda_bwd_ablation.txt 2>&1
python tools/bench_sra.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sra_bench.txt
python tools/bench_nt3.py --shapes all 2>&1 | grep -v amdgpu.ids > gpurun_out/nt3_bench.txt
ls -la gpurun_out/kstats.csv gpurun_out/steady_graph.csv gpurun_out/steady_pvt_ms3_t10.csv gpurun_out/steady_pvt_avss_512.csv gpurun_out/r05_pmc.json gpurun_out/prof_bench_line.json
