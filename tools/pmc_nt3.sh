#!/bin/bash
# PMC counters of the input-gradient GEMM (separate passes per counter set, as MI355X_MICROARCH.md prescribes).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_nt3.txt
: > $OUT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_nt3.py --iters 5 --shapes small --no-lib > /dev/null 2>&1
  python3 - <<PY | tee -a $OUT
import csv,glob,collections
fs=glob.glob("/tmp/pmc/*counter_collection.csv")
if not fs:
    print("no counter file for: $c")
    raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(list))
order=[]
for r in csv.DictReader(open(fs[0])):
    if "gemm_nt3" in r["Kernel_Name"]:
        key=(r["Kernel_Name"][:40], r["Grid_Size"], r.get("LDS_Block_Size",""), r.get("VGPR_Count",""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    n=len(next(iter(v.values())))
    # the three shapes of --shapes small run in order, equally often: report thirds
    for part in range(3):
        print(k, "shape#%d" % part, {c: round(sum(x[part*len(x)//3:(part+1)*len(x)//3])/max(len(x)//3,1),1) for c,x in v.items()}, "n=%d" % (n//3))
PY
  rm -rf /tmp/pmc
done
