#!/bin/bash
# Round-4 visit 14: role-split attn_bwd_dkv_kernel: parity tests, then old vs new timing (interleaved processes, one box)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_train_backward_gpu.py -m gpu -q -x --timeout 900 ) > gpurun_out/v14_pytest_train.log 2>&1; tail -3 gpurun_out/v14_pytest_train.log | cut -c1-300
for rep in 1 2; do
  for v in dkvold dkvnew; do
    BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$v.so BAGEL_ABWD_ONLY=dkv PROBE_LSE=1 timeout 300 python tools/attn_bwd_probe.py 2>&1 | tail -1
  done
done > gpurun_out/v14_dkv_ab.log 2>&1
BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_dkvnew.so BAGEL_ABWD_ONLY=dq PROBE_LSE=1 timeout 300 python tools/attn_bwd_probe.py 2>&1 | tail -1 >> gpurun_out/v14_dkv_ab.log
PROBE_LSE=1 timeout 300 python tools/attn_bwd_probe.py 2>&1 | tail -1 >> gpurun_out/v14_dkv_ab.log
cat gpurun_out/v14_dkv_ab.log
