#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_r3.sh <tag> <variant-to-test-for-parity|-> <variant> ...
# parity run of one variant (full-size + kernel tests), then exp_variants-style timings of all
tag=$1; par=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
if [ "$par" != "-" ]; then
  cp libhm_amd/variants/$par/libhmgpu.so libhm_amd/libhmgpu.so
  timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_streams.py -x -q -m gpu > $out/parity_$par.log 2>&1
  echo "parity $par exit $?"; tail -3 $out/parity_$par.log
  cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
fi
bash tools/exp_variants.sh $tag "$@"
