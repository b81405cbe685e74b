#!/bin/bash
# usage (build container): bash tools/ab_intra.sh  -- libhm_amd/variants/head/libhmgpu.so = the library of the committed sources (the working tree's changes stashed meanwhile)
set -e
cd $(dirname $0)/..
git stash -q
python -c "import libhm_amd.build as b; b.build(verbose=False)" > /dev/null 2>&1
mkdir -p libhm_amd/variants/head && cp libhm_amd/libhmgpu.so libhm_amd/variants/head/libhmgpu.so
git stash pop -q
python -c "import libhm_amd.build as b; b.build(verbose=False)" > /dev/null 2>&1
echo "variants/head built from HEAD; working tree rebuilt"
