#!/bin/bash
# usage (GPU box, repo root): bash tools/traffic_passes.sh <tag>  -- only the FETCH_SIZE / WRITE_SIZE passes of tools/round_profile.sh and their summary
# (gpurun_out/<tag>/hbm_traffic.json): for source changes that leave every kernel's traffic alone but move the digest bench.py checks
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline > $out/pmc_write.log 2>&1
python3 tools/pmc_summary.py --json $out/hbm_traffic.json $out/pmc_fetch $out/pmc_write > $out/pmc_hbm.txt
find $out -name "*.csv" -size +3M -delete
tail -12 $out/pmc_hbm.txt
