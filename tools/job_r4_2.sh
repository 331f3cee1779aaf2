mkdir -p gpurun_out
python tests/graph_compare.py r50 --dump gpurun_out/gc_r50.json > gpurun_out/gc_r50.log 2>&1
python tests/graph_compare.py pvt --dump gpurun_out/gc_pvt.json > gpurun_out/gc_pvt.log 2>&1
python -m pytest tests/test_graph_guard_gpu.py tests/test_dp_gpu.py::test_rccl_all_reduces_the_flat_gradient_buffer_on_one_gpu tests/test_kernels_gpu.py::test_layernorm_fanout_aliases_and_pos_output_match_torch tests/test_kernels_gpu.py::test_deferred_grouped_column_sums_match_immediate_ones tests/test_head_gpu.py tests/test_eval_metric.py tests/test_model_gpu.py::test_configs4_bf16_head_mode_at_full_size tests/test_msda_gpu.py -q -s -m gpu > gpurun_out/t3.log 2>&1; echo rc=$? >> gpurun_out/t3.log
python tools/bench_nt2.py > gpurun_out/nt2_bench.txt 2>&1
for d in 1 2 4 8 16 32 9 13; do echo "== COMBO_NT2_DBG=$d" >> gpurun_out/nt2_abl.txt; COMBO_NT2_DBG=$d python tools/bench_nt2.py --shapes small --no-lib >> gpurun_out/nt2_abl.txt 2>&1; done
bash tools/pmc_nt2.sh > /dev/null 2>&1
tail -5 gpurun_out/t3.log
