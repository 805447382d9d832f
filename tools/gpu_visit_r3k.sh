#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in "" abl2 abl3 abl6 abl10 abl14 abl15; do
  if [ -n "$lib" ]; then export BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$lib.so; else unset BAGEL_HIP_LIB; fi
  echo "== lib=${lib:-product}"
  timeout 200 python tools/attn2_probe.py --iters 20 --only denoise_b8 2>&1 | grep -v amdgpu.ids
done
