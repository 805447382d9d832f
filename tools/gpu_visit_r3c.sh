#!/bin/bash
# round 3, visit C: chunk schedules x fragment prefetch distance of the planned attention kernel, same box
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_attn2_gpu.py -m gpu -q --timeout 600 -x -k "benchmark_launches or deterministic" ) 2>&1 | tail -3
for lib in "" w4; do for s in 0 1 2 3; do
  if [ -n "$lib" ]; then export BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$lib.so; else unset BAGEL_HIP_LIB; fi
  echo "== lib=${lib:-win3} sched=$s"
  BAGEL_ATTN_SCHED=$s timeout 200 python tools/attn2_probe.py --iters 10 --only "$1" 2>&1 | grep -v amdgpu.ids
done; done
