#!/bin/bash
# round 6: same-box A/B of module constants inside the bench step.  usage: tools/job_ab_r6.sh <tag> "<assignments or ->" ...
# every variant runs `bench.py --steps 30 --warmup 5` (graph) once; "-" = the defaults.  Variants are interleaved twice (A B A B).
tag=$1; shift
mkdir -p gpurun_out/r6
B="--no-cpu-baseline --no-other-workloads --steps 30 --warmup 5 $AB_EXTRA"
out=gpurun_out/r6/ab_$tag.txt
rm -f $out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then
    timeout 500 python bench.py $B > gpurun_out/r6/ab_tmp.json 2> gpurun_out/r6/ab_tmp.err
  else
    timeout 500 python tools/ab_const.py $v -- $B > gpurun_out/r6/ab_tmp.json 2> gpurun_out/r6/ab_tmp.err
  fi
  python - "$v" <<'PY' >> $out
import json,sys
try:
    d=json.loads(open("gpurun_out/r6/ab_tmp.json").read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:90s} {d['value']:8.1f} frames/s  {d['ms_per_step']:7.3f} ms  median {d.get('ms_per_step_median')}  {d.get('config',{}).get('launch')}")
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("gpurun_out/r6/ab_tmp.err").read()[-1500:])
PY
done
done
cat $out
