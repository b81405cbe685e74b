#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_dec.sh <tag> <variant> ...   -- the three decode clips with each variant of libhmdec.so (libhm_amd/variants/<v>/libhmdec.so; "base" = as built)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp libhm_amd/libhmdec.so /tmp/libhmdec_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/libhmdec_base.so libhm_amd/libhmdec.so; else cp libhm_amd/variants/$v/libhmdec.so libhm_amd/libhmdec.so; fi
  for s in ra_main10_1920x1080 ldp_main10_3840x2160 ldp_wpp_main10_3840x2160; do
    python3 bench.py --workload decode --stream tests/golden/bench_$s.bin --steps 5 --warmup 1 --no-cpu-baseline > $out/${v}_$s.json 2> $out/${v}_$s.err
    python3 - $out/${v}_$s.json ${v}_$s <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "fps=%.1f check-off=%.1f"%(d["fps"], d["fps_hash_check_off"]), d.get("parse_only_fps"), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
  done
done
cp /tmp/libhmdec_base.so libhm_amd/libhmdec.so
