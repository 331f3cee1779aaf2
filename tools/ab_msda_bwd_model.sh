#!/bin/bash
# the bench step with the two-kernel MSDeformAttn backward (COMBO_MSDA_BWD_WIN=0) and with the fused windowed one (=1)
for w in 0 1; do
  COMBO_MSDA_BWD_WIN=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('COMBO_MSDA_BWD_WIN=$w', d['value'], d['ms_per_step'], {k:(v['avg_launch_us'],v['launches_per_step'],v['frac']) for k,v in d['other_kernels'].items() if 'msda' in k})"
done
