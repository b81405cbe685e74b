#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_sparse.sh <tag>  -- the intra tests of the shipped build, then tools/exp_mixed.sh
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_streams.py tests/test_gpu_gop.py -x -q -m gpu > $out/tests_base.log 2>&1 || { tail -15 $out/tests_base.log; exit 1; }
tail -2 $out/tests_base.log
bash tools/exp_mixed.sh $tag
