#!/bin/bash
# ablation of the fused windowed MSDeformAttn backward: bit 0 = no gather phase, bit 1 = no scatter phase
for d in 0 1 2 3; do
  echo "== COMBO_MSDA_BWD_DBG=$d"
  COMBO_MSDA_BWD_DBG=$d python tools/bench_msda.py --iters 100 2>&1 | grep -E "windowed  "
done
