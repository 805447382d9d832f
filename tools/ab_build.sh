#!/bin/bash
# Build an alternative libbagel_hip (same sources, extra -D flags) next to the product library, for A/B runs on ONE GPU box:
#   bash tools/ab_build.sh alt -DBAGEL_NT_WEIGHTS=0        ->  bagel_amd/libbagel_hip_alt.so
#   BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_alt.so python bench.py ...
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/bagel_amd/csrc/_build_$TAG
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result $*"
pids=()
for f in $ROOT/bagel_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  ( cd $ROOT/bagel_amd/csrc && hipcc $FLAGS -c $f -o $OUT/$b.o 2>/dev/null ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/bagel_amd/libbagel_hip_$TAG.so $OUT/*.o
echo built $ROOT/bagel_amd/libbagel_hip_$TAG.so
