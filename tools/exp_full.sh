#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_full.sh <tag> <variant> ...  -- the default (full) workload with each variant library swapped in
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$v/libhmgpu.so libhm_amd/libhmgpu.so; fi
  python3 bench.py --steps 30 --no-cpu-baseline --no-host-inclusive > $out/$v.json 2> $out/$v.err
  python3 - $out/$v.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels"]
    print(sys.argv[2], "%.1f Gpx/s %.4f ms"%(d["value"]/1000, d["ms_per_step"]), " ".join("%s=%.4f(%.2f)"%(n,v["avg_ms"],v["frac"]) for n,v in k.items()), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
