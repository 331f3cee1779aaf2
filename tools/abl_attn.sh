#!/bin/bash
# forward attention kernel duration (device-side timing slots, back-to-back launches) under the COMBO_ATTN_DBG ablation bits
for d in ${@:-0 16 8 24 25 26 28 31}; do
  echo "dbg=$d $(COMBO_ATTN_DBG=$d timeout 60 python3 tools/bench_attn.py 784 2>&1 | grep 'device-side: back' | sed 's/\[attention Lk=784\]//')"
done
