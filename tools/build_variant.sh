#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]  -- libhmgpu.so built with extra flags into libhm_amd/variants/<name>/ (experiments;
# tools/exp_variants.sh swaps it in on the GPU box).  Only the MC kernels are recompiled, the other objects come from libhm_amd/build/.
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
d=$root/libhm_amd/variants/$name
mkdir -p $d
ARCH="--offload-arch=gfx950:xnack-"      # as libhm_amd/build.py
FLAGS="$ARCH -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value"
for f in ${VARIANT_SOURCES:-k_mc.hip}; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $root/libhm_amd/csrc/$f -o $d/${f%.hip}.o &
done
wait
objs=""
for o in $root/libhm_amd/build/*.o; do
  b=$(basename $o)
  if [ -f $d/$b ]; then objs="$objs $d/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc $ARCH -shared -fPIC -o $d/libhmgpu.so $objs
echo $d/libhmgpu.so
