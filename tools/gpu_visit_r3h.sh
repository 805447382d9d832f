#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_attn2_gpu.py -m gpu -q --timeout 600 -x ) 2>&1 | tail -2
for s in 0 1 0 1; do
  echo "== sched=$s"
  BAGEL_ATTN_SCHED=$s timeout 200 python tools/attn2_probe.py --iters 20 --only "$1" 2>&1 | grep -v amdgpu.ids
done
