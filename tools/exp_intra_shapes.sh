#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_intra_shapes.sh  -- k_intra on all-intra pictures of several shapes (one CTU row, one / two CTU columns, full sizes): the lags of the CTU wavefront
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IFS=","; for wh in ${SHAPES:-3840 64,3840 128,64 2160,128 2160,1920 1088,3840 2160}; do
  IFS=" "; set -- $wh
  timeout -k 10 120 python3 bench.py --workload intra --batch 1 --steps 5 --warmup 1 --profile-steps 2 --no-cpu-baseline --width $1 --height $2 > gpurun_out/shape.json 2> gpurun_out/shape.err
  python3 - "$1 x $2" <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/shape.json")); print(sys.argv[1], "intra=%.3f ms"%d["kernels"]["intra"]["avg_ms"], flush=True)
except Exception as e: print(sys.argv[1], "FAILED", e, flush=True)
PY
done
