#!/bin/bash
# Attention reverse: parity tests of the product library, then per-kernel timing of libbagel_hip_dkvold.so (round 3's kernels) vs libbagel_hip_dkvnew.so
# (this tree, built with -DBAGEL_ENABLE_ABLATIONS so that BAGEL_ABWD_ONLY works), interleaved processes on one box.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_train_backward_gpu.py -m gpu -q -x --timeout 900 ) > gpurun_out/bwd_pytest_train.log 2>&1; tail -3 gpurun_out/bwd_pytest_train.log | cut -c1-300
for rep in 1 2; do
  for v in dkvold dkvnew; do
    for only in dkv dq; do
      BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$v.so BAGEL_ABWD_ONLY=$only PROBE_LSE=1 timeout 300 python tools/attn_bwd_probe.py 2>&1 | tail -1
    done
  done
done > gpurun_out/bwd_ab.log 2>&1
PROBE_LSE=1 timeout 300 python tools/attn_bwd_probe.py 2>&1 | tail -1 >> gpurun_out/bwd_ab.log
cat gpurun_out/bwd_ab.log
