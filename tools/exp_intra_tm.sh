#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_intra_tm.sh <tag> [bench.py arguments]  -- one run with the INTRA_TIMING build (libhm_amd/variants/tm):
# per-CTU phase times of k_intra on stdout (default: the intra workload, one picture)
tag=$1; shift
args=${@:---workload intra --batch 1}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
cp libhm_amd/variants/tm/libhmgpu.so libhm_amd/libhmgpu.so
timeout -k 10 200 python3 bench.py $args --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-host-inclusive > $out/tm.out 2> $out/tm.err
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
grep -c "^TM" $out/tm.out
