#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_write_probe.sh <outdir> <bench args...>  -- WRITE_SIZE / FETCH_SIZE per kernel for a bench variant
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline "$@" > $out/w.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline "$@" > $out/f.log 2>&1
python3 tools/pmc_summary.py $out/w $out/f
