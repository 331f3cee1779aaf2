mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_notiming.json 2> gpurun_out/bench_notiming.err
python bench.py --no-cpu-baseline > gpurun_out/bench_timing.json 2> gpurun_out/bench_timing.err
python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_notiming2.json 2> gpurun_out/bench_notiming2.err
python - <<'PY'
import json
for f in ("bench_notiming","bench_timing","bench_notiming2"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["ms_per_step_median"])
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-800:])
PY
