mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke_rc=$? >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
s=$(date +%s); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; e=$(date +%s); echo "driver-like bench wall: $((e-s)) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("other_workloads"), d.get("cpu_baseline"))
PY
python bench.py --no-cpu-baseline --dump-slots 2 > gpurun_out/bench_slots.json 2> gpurun_out/bench_slots.err
python bench.py --mode infer --no-cpu-baseline 2>/dev/null | cut -c1-300
python bench.py --head-dtype bf16 --no-cpu-baseline 2>/dev/null | cut -c1-200
python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | cut -c1-200
python bench.py --config pvt_s4 --no-cpu-baseline 2>/dev/null | cut -c1-200
