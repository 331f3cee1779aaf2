#!/bin/bash
# per-kernel register / spill / LDS / occupancy table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage)
src=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/root/repo/combo-avs_amd/csrc -I/root/repo/include -c $src -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r"remark: +([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)",l)
    if m and cur: rows[cur][m.group(1).strip()]=m.group(2)
import subprocess
for k,v in rows.items():
    name=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip()[:140]
    print(name); print("    ", {a:b for a,b in v.items() if a in ("VGPRs","AGPRs","SGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","SGPRs Spill","VGPRs Spill","LDS Size [bytes/block]")})
'
rm -f /tmp/kr_$$.o
