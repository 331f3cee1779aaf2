mkdir -p gpurun_out
python tests/graph_compare.py pvt --frozen --dump gpurun_out/gc_pvt_frozen.json > gpurun_out/gc_pvt_frozen.log 2>&1
python -m pytest tests/test_graph_gpu.py tests/test_head_gpu.py tests/test_model_gpu.py::test_configs4_bf16_head_mode_at_full_size -q -s -m gpu > gpurun_out/t4.log 2>&1; echo rc=$? >> gpurun_out/t4.log
python tools/bench_nt2.py --shapes big > gpurun_out/nt2_bench.txt 2>&1
rm -f gpurun_out/nt2_abl.txt
for d in 1 2 4 8 16 32 9 41 43 47 63; do echo "== COMBO_NT2_DBG=$d" >> gpurun_out/nt2_abl.txt; COMBO_NT2_DBG=$d python tools/bench_nt2.py --shapes small --no-lib >> gpurun_out/nt2_abl.txt 2>&1; done
bash tools/pmc_nt2.sh > /dev/null 2>&1
tail -5 gpurun_out/t4.log
