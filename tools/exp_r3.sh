#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_r3.sh <tag> "<variants to test for parity, or ->" <variant> ...
# parity run of the named variants (MC-related GPU tests), then exp_variants-style timings of all
tag=$1; par=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
if [ "$par" != "-" ]; then
  for p in $par; do
    if [ $p = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$p/libhmgpu.so libhm_amd/libhmgpu.so; fi
    timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_streams.py tests/test_gpu_gop.py -x -q -m gpu > $out/parity_$p.log 2>&1
    echo "parity $p exit $?"; tail -3 $out/parity_$p.log
  done
  cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
fi
bash tools/exp_variants.sh $tag "$@"
