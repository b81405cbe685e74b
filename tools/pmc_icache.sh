#!/bin/bash
# usage (GPU box, repo root): BENCH_ARGS="--workload intra --batch 1" bash tools/pmc_icache.sh <outdir>  -- instruction-cache counters of a short bench run
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $out/ic -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline $BENCH_ARGS > $out/ic.log 2>&1 || echo "icache pass failed"
python3 tools/pmc_kernel_sums.py $out > $out/sums.txt 2>&1
tail -40 $out/sums.txt
