#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_attn2_gpu.py -m gpu -q --timeout 600 -x ) 2>&1 | tail -3
for r in 1 2; do for lib in "" s3; do
  if [ -n "$lib" ]; then export BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$lib.so; else unset BAGEL_HIP_LIB; fi
  echo "== lib=${lib:-slots4}"
  timeout 200 python tools/attn2_probe.py --iters 20 2>&1 | grep -v amdgpu.ids
done; done
