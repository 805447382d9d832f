#!/bin/bash
# round 6, visit 14: 32-request decode, RMSNorm fused into the two-block weight stream vs one RMSNorm launch in front of it (same box, alternating)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_decode_gpu.py -x -q ) > gpurun_out/v14_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/v14_rc.txt
for k in 0 1 0 1; do
  ( BAGEL_MB2_FUSED_NORM=$k timeout 900 python bench.py --gpus 1 --only-understanding --no-cpu-baseline --no-int8 ) > gpurun_out/v14_und_fused$k.log 2>> gpurun_out/v14_und.err
  echo "fused=$k $(grep -o '"batched_decode_32": {[^}]*' gpurun_out/v14_und_fused$k.log | cut -c1-160)" >> gpurun_out/v14_ab.log
done
cat gpurun_out/v14_rc.txt; tail -2 gpurun_out/v14_tests.log; cat gpurun_out/v14_ab.log
