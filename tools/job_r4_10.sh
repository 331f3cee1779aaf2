mkdir -p gpurun_out
python tools/ab_const.py combo_avs_amd.ops.convwrw.DX_OWN=3 -- --no-cpu-baseline > gpurun_out/bench_dx3.json 2> gpurun_out/bench_dx3.err
python bench.py --no-cpu-baseline > gpurun_out/bench_dx2.json 2> gpurun_out/bench_dx2.err
python - <<'PY'
import json
for f in ("bench_dx3","bench_dx2"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        k=d["other_kernels"].get("gemm_nt3_kernel") or {}
        print(f, d["value"], d["ms_per_step"], "x3:", k.get("ms_per_step"), k.get("frac"), k.get("launches_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
