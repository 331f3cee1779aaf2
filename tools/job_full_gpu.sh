mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/t_full.log 2>&1; echo rc=$? >> gpurun_out/t_full.log
tail -6 gpurun_out/t_full.log
