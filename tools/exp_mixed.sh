#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_mixed.sh <tag>  -- k_intra on mixed P pictures (5 / 10 / 25 % intra CUs, 16 pictures) and on I pictures (1 and 16)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 0.05 0.1 0.25; do
  python3 bench.py --intra-frac $f --steps 10 --no-cpu-baseline --no-host-inclusive > $out/mixed_$f.json 2> $out/mixed_$f.err
  python3 -c "
import json; d=json.load(open('$out/mixed_$f.json')); print('intra-frac $f', d['value'], 'Mpx/s', 'intra=%.3f ms' % d['kernels']['intra']['avg_ms'], 'filter=%.3f' % d['kernels']['filter_fused']['avg_ms'], flush=True)"
done
for b in 1 16; do
  python3 bench.py --workload intra --batch $b --steps 5 --warmup 1 --profile-steps 2 --no-cpu-baseline > $out/intra_$b.json 2> $out/intra_$b.err
  python3 -c "
import json; d=json.load(open('$out/intra_$b.json')); print('I pictures x$b', 'intra=%.3f ms' % d['kernels']['intra']['avg_ms'], flush=True)"
done
