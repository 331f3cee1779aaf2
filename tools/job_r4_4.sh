mkdir -p gpurun_out
python -m pytest tests -q -x -m gpu > gpurun_out/t5.log 2>&1; echo rc=$? >> gpurun_out/t5.log
python bench.py --no-cpu-baseline --dump-slots 2 > gpurun_out/bench_slots.json 2> gpurun_out/bench_slots.err
tail -4 gpurun_out/t5.log
