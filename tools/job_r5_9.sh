#!/bin/bash
for mode in same images instances masks_only classes_only; do
  timeout 300 python tools/dbg_avss_graph2.py $mode 2>&1 | grep -v "amdgpu.ids\|Cannot find the function" | tail -3
done
