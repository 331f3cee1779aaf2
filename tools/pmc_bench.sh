#!/bin/bash
# HBM traffic (and MFMA / LDS counters) of the head's instrumented kernels inside the BENCH step, per launch, as
# MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit together), kernel trace
# only; FETCH_SIZE is doubled (gfx950 counts 128-B requests of wide coalesced reads at 64 B).  The step runs eagerly
# (--no-graph) so that every launch is its own dispatch record.  Writes gpurun_out/r06_pmc.json, stamped with $COMBO_COMMIT;
# copy it to profiles/ - bench.py reports its numbers in `roofline.traffic`.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench.txt
: > $OUT
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  COMBO_MIOPEN_BENCHMARK=0 COMBO_GEMM_TUNING=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
  python3 - <<PY | tee -a $OUT
import csv,glob,collections,re
fs=glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
if fs:
    for r in csv.DictReader(open(fs[0])):
        n=r["Kernel_Name"]
        m=re.search(r"(gemm_nt_f32_kernel|gemm_smallm_kernel|gemm_nt3_kernel|gemm_tn_grouped_kernel|gemm_tn_glds_kernel|conv3x3_wgrad_kernel|attn_fwd_kernel|attn_bwd_dq_kernel|attn_bwd_dkv_kernel|msda_fwd_tap_d32|msda_bwd_\w+|bifuse_\w+|add_ln_fwd_kernel|ln_bwd_kernel|presplit_kernel)", n)
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print("PMC", k, {c: [round(sum(x)/len(x),1), len(x)] for c,x in v.items()})
PY
  rm -rf /tmp/pmc
done
python3 - <<PY
import ast, json, os, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
out={"commit": os.environ.get("COMBO_COMMIT","unknown"), "csrc_sha256": bench.csrc_digest(), "frames_per_launch": 40,
     "command": "rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --no-graph --steps 2 --warmup 1 (tools/pmc_bench.sh); FETCH_SIZE x 2 (gfx950), KiB"}
for line in open("$OUT"):
    if not line.startswith("PMC "): continue
    _, name, rest = line.split(" ", 2)
    d=ast.literal_eval(rest)
    rec=out.setdefault(name, {})
    for c,(avg,n) in d.items():
        rec[c]=avg; rec["launches_profiled"]=n
for name,rec in out.items():
    if isinstance(rec, dict) and "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_bytes_per_launch"]=int((2.0*rec["FETCH_SIZE"]+rec["WRITE_SIZE"])*1024)
json.dump(out, open(os.path.join("$GRAFT_REPO_ROOT","gpurun_out","r06_pmc.json"),"w"), indent=1)
print(json.dumps({k:(v.get("hbm_bytes_per_launch") if isinstance(v,dict) else v) for k,v in out.items()}))
PY
