#!/bin/bash
# usage (on the GPU box, repo root): bash tools/round_profile.sh <tag>   -- bench JSONs, rocprofv3 kernel stats and HBM-byte
# PMC passes of the default bench command, written under gpurun_out/<tag>/ (copy what is to be kept into profiles/)
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench_full.json 2> $out/bench_full.err
for wl in idct mc mc_bi filter; do
  python3 bench.py --workload $wl --steps 10 --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err
done
python3 bench.py --workload intra --batch 1 --steps 5 --warmup 1 --profile-steps 2 --no-cpu-baseline > $out/bench_intra_1pic.json 2> $out/bench_intra_1pic.err
python3 bench.py --workload intra --steps 5 --warmup 1 --profile-steps 2 --no-cpu-baseline > $out/bench_intra.json 2> $out/bench_intra.err
# P pictures with a realistic share of intra CUs reconstructed on the device (intra modes supplied): what the wavefront costs in mixed pictures
for f in 0.05 0.1 0.25; do
  python3 bench.py --intra-frac $f --steps 10 --no-cpu-baseline --no-host-inclusive > $out/bench_full_intra_$f.json 2> $out/bench_full_intra_$f.err
done
python3 bench.py --workload gop --steps 10 > $out/bench_gop.json 2> $out/bench_gop.err
python3 bench.py --streams 1 --no-cpu-baseline > $out/bench_full_1lane.json 2> $out/bench_full_1lane.err
python3 bench.py --batch 8 --no-cpu-baseline > $out/bench_full_batch8.json 2> $out/bench_full_batch8.err
python3 bench.py --workload decode --stream tests/golden/bench_ldp_main10_3840x2160.bin --steps 3 --warmup 1 > $out/bench_decode_2160p.json 2> $out/bench_decode_2160p.err
python3 bench.py --workload decode --stream tests/golden/bench_ra_main10_1920x1080.bin --steps 5 --warmup 2 > $out/bench_decode_1080p.json 2> $out/bench_decode_1080p.err
python3 bench.py --workload decode --steps 3 --warmup 1 > $out/bench_decode_2160p_wpp.json 2> $out/bench_decode_2160p_wpp.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline > $out/pmc_write.log 2>&1
python3 tools/trace_summary.py $(find $out/stats -name "*kernel_trace.csv" | head -1) > $out/kernel_stats_batched.csv
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 tools/pmc_summary.py --json $out/hbm_traffic.json $out/pmc_fetch $out/pmc_write > $out/pmc_hbm.txt
find $out -name "*.csv" -size +3M -delete      # per-dispatch traces are large; the stats summaries stay
ls -R $out | head -40
