#!/bin/bash
# tests of the fused mask path + A/B of the bench step with / without it
python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_mask or mask_logits_of_all" 2>&1 | tail -8
python -m pytest tests/test_head_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_dp_gpu.py -q -x 2>&1 | tail -5
for f in 0 1; do
  COMBO_FUSED_MASKS=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('COMBO_FUSED_MASKS=$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches_per_step'])"
done
