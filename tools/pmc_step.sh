#!/bin/bash
# PMC counters of the head's own kernels inside one eager training step (separate passes per counter set, no tracing
# domains besides the kernel trace): VALU / MFMA / LDS instruction counts and busy cycles per kernel.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step.txt
: > $OUT
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | tr '\n' ' ' | tee -a $OUT; echo | tee -a $OUT
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU"; do
  COMBO_MIOPEN_BENCHMARK=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY | tee -a $OUT
import csv,glob,collections,re
fs=glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    m=re.search(r"(gemm_nt2_kernel<\w+|gemm_tn_grouped_kernel|conv3x3_wgrad_kernel|msda_\w+|bifuse_\w+|splitk_reduce_grouped_kernel|gn_\w+|matcher_cost_kernel|mask_loss_\w+|cosine_\w+|adamw_kernel|relu_grad\w*|bias_act_kernel|presplit_kernel|lngrad\w*|ln_\w+)", n)
    if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
  rm -rf /tmp/pmc
done
