#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_full2.sh <tag> <variant> ...  -- default bench (per-kernel times) + the intra workload with each variant library
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$v/libhmgpu.so libhm_amd/libhmgpu.so; fi
  python3 bench.py --no-cpu-baseline > $out/${v}_full.json 2> $out/${v}_full.err
  python3 bench.py --workload intra --batch 1 --steps 5 --warmup 1 --profile-steps 2 --no-cpu-baseline > $out/${v}_intra.json 2> $out/${v}_intra.err
  python3 - $out/${v}_full.json $out/${v}_intra.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels"]; i=json.load(open(sys.argv[2]))
    print(sys.argv[3], "value=%.0f ms=%.4f"%(d["value"], d["ms_per_step"]), " ".join("%s=%.4f"%(n,k[n]["avg_ms"]) for n in k), "intra=%.3f"%i["kernels"]["intra"]["avg_ms"], flush=True)
except Exception as e:
    print(sys.argv[3], "FAILED", e, flush=True)
PY
done
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
