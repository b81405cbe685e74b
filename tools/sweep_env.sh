#!/bin/bash
# usage: bash tools/sweep_env.sh VAR kernel v1 v2 ...   -- bench once per value of an environment tuning knob
var=$1; kern=$2; shift 2
for v in "$@"; do
  export $var=$v
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > /tmp/sweep.json
  python -c "import json; d=json.load(open('/tmp/sweep.json')); print('$var=$v', d['value'], d['ms_per_step'], d['kernels']['$kern'])"
done
