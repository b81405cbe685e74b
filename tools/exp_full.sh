#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_full.sh <tag> <variant> ...   -- the default workload's per-kernel times with each variant
# library of libhm_amd/variants/ swapped in ("base" = the library as built); EXP_ARGS = extra bench.py arguments
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$v/libhmgpu.so libhm_amd/libhmgpu.so; fi
  python3 bench.py --steps 20 --no-cpu-baseline --no-host-inclusive $EXP_ARGS > $out/$v.json 2> $out/$v.err
  python3 - $out/$v.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels"]
    print(sys.argv[2], "value=%.1f"%d["value"], " ".join("%s=%.4f(%.2f)"%(n,k[n]["avg_ms"],k[n]["frac"]) for n in k), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
