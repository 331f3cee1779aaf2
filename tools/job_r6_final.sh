#!/bin/bash
# round 6, final evidence at one commit: PMC traffic first (bench.py reads profiles/r06_pmc.json of the same csrc digest), the whole
# GPU test suite, the default bench run, the rocprofv3 kernel trace + steady-state summary, the PVT workloads.  -> gpurun_out/
mkdir -p gpurun_out profiles
export COMBO_COMMIT=$(git rev-parse --short HEAD 2>/dev/null || cat .combo_commit 2>/dev/null || echo unknown)
if [ "$1" != "--no-pmc" ]; then
  bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
  cp gpurun_out/r06_pmc.json profiles/r06_pmc.json 2>/dev/null   # (on the box: the bench below then reports `traffic`)
fi
if [ "$2" != "--no-tests" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_tests.log 2>&1
  echo "tests exit $?" >> gpurun_out/final_tests.log
  tail -3 gpurun_out/final_tests.log
fi
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -c 400 gpurun_out/final_bench.json
bash tools/final_profile.sh --profile-only > gpurun_out/final_profile.log 2>&1
python bench.py --single-stream --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --dump-slots 2 > gpurun_out/slots_single.json 2> gpurun_out/slots_single.err
ls -la gpurun_out/kstats.csv gpurun_out/steady_graph.csv gpurun_out/r06_pmc.json gpurun_out/prof_bench_line.json
bash tools/prof_config.sh pvt_ms3_t10 > gpurun_out/prof_ms3.log 2>&1
bash tools/prof_config.sh pvt_avss_512 3 2 > gpurun_out/prof_avss.log 2>&1
ls -la gpurun_out/steady_pvt_ms3_t10.csv gpurun_out/steady_pvt_avss_512.csv
