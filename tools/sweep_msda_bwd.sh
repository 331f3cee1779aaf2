#!/bin/bash
# A/B of the fused windowed MSDeformAttn backward: waves per workgroup x LDS cap per workgroup (KB) at the bench shape
for cfg in "12 160" "6 76" "8 76" "4 50" "6 50" "4 38"; do
  set -- $cfg
  echo "== waves $1, LDS cap $2 KB"
  COMBO_MSDA_BWD_WAVES=$1 COMBO_MSDA_BWD_LDS_KB=$2 python tools/bench_msda.py --iters 100 2>&1 | grep -E "windowed  "
done
