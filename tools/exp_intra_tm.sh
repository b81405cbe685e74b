#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_intra_tm.sh <tag>  -- one run of the intra workload with the INTRA_TIMING build (libhm_amd/variants/tm): per-CTU phase times on stdout
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
cp libhm_amd/variants/tm/libhmgpu.so libhm_amd/libhmgpu.so
timeout -k 10 200 python3 bench.py --workload intra --batch 1 --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > $out/tm.out 2> $out/tm.err
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
grep -c "^TM" $out/tm.out
