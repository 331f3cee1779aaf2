mkdir -p gpurun_out
python bench.py --no-cpu-baseline --dump-slots 1,3,4,5 > gpurun_out/bench_slots13.json 2> gpurun_out/bench_slots13.err
tail -1 gpurun_out/bench_slots13.json | cut -c1-200
