#!/bin/bash
# Timing ablations of the attention reverse on ONE GPU box: one library per AB_ABL value (only attention_bwd.o differs), each timed on
# the dq kernel, the dkv kernel and both (tools/attn_bwd_probe.py).  Build the variants in the build container first: bash tools/ab_attn_bwd.sh build
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VARIANTS="0 1 2 4 8 16 28 32 64 96"
if [ "$1" = "build" ]; then
  for v in $VARIANTS; do
    mkdir -p $ROOT/bagel_amd/csrc/_build_abl
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAB_ABL=$v -DBAGEL_ENABLE_ABLATIONS -c $ROOT/bagel_amd/csrc/attention_bwd.hip -o $ROOT/bagel_amd/csrc/_build_abl/attention_bwd_$v.o &
  done
  wait
  for v in $VARIANTS; do
    objs=$(ls $ROOT/bagel_amd/csrc/_build/*.o | grep -v attention_bwd.o)
    hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/bagel_amd/libbagel_hip_abl$v.so $objs $ROOT/bagel_amd/csrc/_build_abl/attention_bwd_$v.o
  done
  ls -la $ROOT/bagel_amd/libbagel_hip_abl*.so | wc -l
  exit 0
fi
for v in $VARIANTS; do
  for only in dq dkv; do
    BAGEL_HIP_LIB=$ROOT/bagel_amd/libbagel_hip_abl$v.so BAGEL_ABWD_ONLY=$only python $ROOT/tools/attn_bwd_probe.py 2>&1 | tail -1
  done
done
