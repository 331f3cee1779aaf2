#!/bin/bash
COMBO_TEST_VERBOSE=1 python -m pytest tests/test_head_gpu.py -q -x -s -k "bf16_forward_mode" 2>&1 | grep -E "bf16 mode|passed|failed|assert|Error" | tail -8
for hd in fp32 bf16; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --head-dtype $hd 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('--head-dtype $hd', d['value'], d['ms_per_step'], d['dtype'], {k:(v['ms_per_step'],v['frac']) for k,v in list(d['other_kernels'].items())[:3]}, d['roofline']['kernel'], d['roofline']['ms_per_step'])"
done
