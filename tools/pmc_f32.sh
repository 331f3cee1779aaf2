#!/bin/bash
# PMC counters of the exact-fp32 GEMM kernel (separate passes per counter set, as MI355X_MICROARCH.md prescribes).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_f32.txt
: > $OUT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_f32.py --iters 5 --shapes small --no-lib > /dev/null 2>&1
  python3 - <<PY | tee -a $OUT
import csv,glob,collections,re
fs=glob.glob("/tmp/pmc/*counter_collection.csv")
if not fs:
    print("no counter file for: $c")
    raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "gemm_nt_f32" in r["Kernel_Name"]:
        key=(r["Grid_Size"], r.get("LDS_Block_Size",""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print("gemm_nt_f32 grid/lds", k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
  rm -rf /tmp/pmc
done
