#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_mixed_variants.sh <tag> <variant|base> ...  -- the mixed-picture lines (5 / 10 / 25 % intra CUs) with each library
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$v/libhmgpu.so libhm_amd/libhmgpu.so; fi
  for f in 0.05 0.1 0.25; do
    timeout -k 10 200 python3 bench.py --intra-frac $f --steps 10 --no-cpu-baseline --no-host-inclusive > $out/${v}_$f.json 2> $out/${v}_$f.err
    python3 -c "
import json; d=json.load(open('$out/${v}_$f.json')); print('$v intra-frac $f', d['value'], 'Mpx/s', 'intra=%.3f ms' % d['kernels']['intra']['avg_ms'], flush=True)"
  done
done
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
