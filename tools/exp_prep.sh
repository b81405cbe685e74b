#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_prep.sh <tag> <variant> ...  -- k_prep time on the default mix with each variant library
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$v/libhmgpu.so libhm_amd/libhmgpu.so; fi
  python3 bench.py --workload mc --steps 10 --no-cpu-baseline > $out/${v}.json 2> $out/${v}.err
  python3 - $out/${v}.json ${v} <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels"]
    print(sys.argv[2], " ".join("%s=%.4f"%(n,k[n]["avg_ms"]) for n in ("prep","mc_luma","itx") if n in k), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
