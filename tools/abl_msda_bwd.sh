#!/bin/bash
# Ablation of the fused windowed MSDeformAttn backward (csrc/msda_bwd.hip, COMBO_MSDA_BWD_DBG bits; results are wrong with any
# bit set - timing only): 1 = no gather phase (grad_loc / grad_w), 2 = no scatter phase (grad_value), 16 = no grad_out scan
# (prepass), 32 = no pass 0 (row scales), 64 = no main loop, 128 = no slab staging.
for d in 0 1 2 3 16 32 48 64 128; do
  echo "== COMBO_MSDA_BWD_DBG=$d"
  COMBO_MSDA_BWD_DBG=$d python tools/bench_msda.py --iters 100 2>&1 | grep -E "windowed  "
done
