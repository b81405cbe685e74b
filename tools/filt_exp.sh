set -e
mkdir -p gpurun_out
for v in 0 7; do
  echo "== var $v" >> gpurun_out/filt_exp.log
  HMGPU_FILT_VAR=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 >> gpurun_out/filt_exp.log 2>&1
done
