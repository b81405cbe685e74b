#!/bin/bash
# usage: tools/kres.sh <file.hip> -- VGPRs / SGPRs / scratch / occupancy / LDS of every kernel in the file (gfx950)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r"remark: (.*?)(?: \[-Rpass)",l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        cur=t.split(":",1)[1].strip(); print(); print(cur[:70],end=" | ")
    elif any(t.startswith(k) for k in ("VGPRs:","TotalSGPRs","ScratchSize","Occupancy","LDS Size")):
        print(t.replace(" [bytes/lane]","").replace(" [waves/SIMD]","").replace(" [bytes/block]",""),end=" | ")
print()'
