#!/bin/bash
# same-box A/B of the round-5 backbone switches (three runs each way, interleaved)
mkdir -p gpurun_out
: > gpurun_out/r5_14_ab.txt
for rep in 1 2; do
  for mode in on off; do
    if [ $mode = on ]; then sw=""; else sw="combo_avs_amd.ops.convwrw.WGRAD_ANY_C=0 combo_avs_amd.ops.convwrw.WGRAD_S2=0 combo_avs_amd.ops.convwrw.DX_S2_1X1=0"; fi
    python tools/ab_const.py $sw -- --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j = json.loads(l); print('$mode rep $rep', j['value'], 'frames/s', j['ms_per_step'], 'ms per step')
" >> gpurun_out/r5_14_ab.txt
  done
done
cat gpurun_out/r5_14_ab.txt
