mkdir -p gpurun_out
python bench.py --no-cpu-baseline --dump-slots 2 > gpurun_out/bench_nt3.json 2> gpurun_out/bench_nt3.err
COMBO_DX_KERNEL=2 python bench.py --no-cpu-baseline > gpurun_out/bench_nt2.json 2> gpurun_out/bench_nt2.err
python - <<'PY'
import json
for f in ("bench_nt3","bench_nt2"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        k=d["other_kernels"].get("gemm_nt2_kernel") or {}
        print(f, d["value"], d["ms_per_step"], "x3:", k.get("ms_per_step"), k.get("frac"), k.get("launches_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
