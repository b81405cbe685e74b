#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_mc.sh <tag>  -- MC kernel times by PU size / MV range + PMC passes of the mc workload
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python3 bench.py --workload mc "$@" --steps 10 --no-cpu-baseline > $out/$name.json 2> $out/$name.err; python3 - $out/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[2], " ".join("%s=%.4f(%.2f)"%(n,k[n]["avg_ms"],k[n]["frac"]) for n in ("prep","mc_luma","mc_chroma","itx") if n in k))
PY
}
run mix
run p64 --mode-probs 1,0,0,0,0
run p32 --mode-probs 0,1,0,0,0
run p16 --mode-probs 0,0,1,0,0
run p8 --mode-probs 0,0,0,1,0
run amp --mode-probs 0,0,0,0,1
run mv0 --mv-range 0
run mv8 --mv-range 8
run bi --workload mc_bi
