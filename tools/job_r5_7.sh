#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_sra_gpu.py -q -m gpu -x > gpurun_out/r5_7_tests_sra.log 2>&1
echo "tests rc $?" >> gpurun_out/r5_7_tests_sra.log
python tools/bench_sra.py > gpurun_out/r5_7_sra_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o ps --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_sra.py > /tmp/ps.log 2>&1 < /dev/null
f=$(find /tmp/ps -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python3 - "$f" > gpurun_out/r5_7_sra_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per kernel name + grid: list durations in order
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("sra_", "attn_fwd", "bwd_kernel")):
        continue
    key = (n[:70], r["Grid_Size_X"], r.get("Workgroup_Size_X", ""))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v2 = sorted(v)
    print(f"{k[0]:70s} grid {k[1]:>9s} wg {k[2]:>4s} calls {len(v):4d} median_us {v2[len(v2)//2]:8.1f} min {v2[0]:8.1f}")
PY
tail -3 gpurun_out/r5_7_tests_sra.log; cat gpurun_out/r5_7_sra_bench.txt; cat gpurun_out/r5_7_sra_kernels.txt
