mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke_rc=$? >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "edges" 2>&1 | tail -2
s=$(date +%s); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; e=$(date +%s); echo "driver-like bench wall: $((e-s)) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("other_workloads"), d.get("cpu_baseline"))
PY
