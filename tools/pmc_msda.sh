#!/bin/bash
# PMC counters of the MSDeformAttn kernels (separate passes: FETCH_SIZE / WRITE_SIZE do not fit together with SQ sets).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_msda.txt
: > $OUT
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAVES"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_msda.py --iters 3 > /dev/null 2>&1
  python3 - <<PY | tee -a $OUT
import csv,glob,collections,re
f=glob.glob("/tmp/pmc/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m=re.search(r"msda_\w+", r["Kernel_Name"])
    if m and ("lds" in m.group(0) or "tap" in m.group(0)): agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
  rm -rf /tmp/pmc
done
