#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_variants.sh <tag> <variant> ...   -- runs the mc workload (default mix, 64x64, 8x8) with each
# variant library of libhm_amd/variants/ swapped in ("base" = the library as built)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmgpu.so /tmp/libhmgpu_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so; else cp libhm_amd/variants/$v/libhmgpu.so libhm_amd/libhmgpu.so; fi
  for cfg in "mix" "p64 --mode-probs 1,0,0,0,0" "p32 --mode-probs 0,1,0,0,0" "p16 --mode-probs 0,0,1,0,0" "p8 --mode-probs 0,0,0,1,0" "bi --workload mc_bi"; do
    set -- $cfg; name=$1; shift
    python3 bench.py --workload mc "$@" --steps 10 --no-cpu-baseline > $out/${v}_$name.json 2> $out/${v}_$name.err
    python3 - $out/${v}_$name.json ${v}_$name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels"]
    print(sys.argv[2], " ".join("%s=%.4f(%.2f)"%(n,k[n]["avg_ms"],k[n]["frac"]) for n in ("prep","mc_luma","mc_chroma","itx") if n in k), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
  done
done
cp /tmp/libhmgpu_base.so libhm_amd/libhmgpu.so
