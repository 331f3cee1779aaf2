#!/bin/bash
# fused windowed MSDeformAttn backward (csrc/msda_bwd.hip): LDS cap per workgroup (KB) -> window table; ablation bits
# (COMBO_MSDA_BWD_DBG: 1 = no gather phase, 2 = no scatter phase, 16 = no grad_out scan, 32 = no pass 0, 64 = no main loop,
# 128 = no slab staging; 67 = 3 + 64 ...) at the bench shape (tools/bench_msda.py: random reference
# points, i.e. NO locality between consecutive queries - the model's raster-ordered queries let bands skip most iterations)
for cfg in "8 80" "16 160" "16 120" "8 160"; do
  set -- $cfg
  echo "== waves $1, LDS cap $2 KB"
  COMBO_MSDA_BWD_WAVES=$1 COMBO_MSDA_BWD_LDS_KB=$2 python tools/bench_msda.py --iters 100 2>&1 | grep -E "windowed  |two-kernel"
done
for d in 3; do
  echo "== COMBO_MSDA_BWD_DBG=$d (default cap)"
  COMBO_MSDA_BWD_DBG=$d python tools/bench_msda.py --iters 100 2>&1 | grep -E "windowed  "
done
