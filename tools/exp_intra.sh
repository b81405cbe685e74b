#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_intra.sh <tag> <variant> ...   -- the intra workload (1 and 16 pictures) with each variant library
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$v/libhmgpu.so libhm_amd/libhmgpu.so; fi
  for b in 1 16; do
    timeout -k 10 200 python3 bench.py --workload intra --batch $b --steps 5 --warmup 1 --profile-steps 2 --no-cpu-baseline > $out/${v}_$b.json 2> $out/${v}_$b.err
    python3 - $out/${v}_$b.json ${v}_$b <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "intra=%.3f ms"%d["kernels"]["intra"]["avg_ms"], flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
  done
done
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
