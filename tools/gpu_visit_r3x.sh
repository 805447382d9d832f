#!/bin/bash
# round 3, visit x: dq kernel at two waves per SIMD: parity, probe with and without forward statistics, 28-layer step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_backward_gpu.py tests/test_train_gpu.py -q 2>&1 | tail -4 > gpurun_out/r3x_pytest.log
tail -4 gpurun_out/r3x_pytest.log
python tools/attn_bwd_probe.py 2>&1 | tail -1
PROBE_LSE=1 python tools/attn_bwd_probe.py 2>&1 | tail -1
PROBE_ITERS=2 timeout 900 python tools/train_step_probe.py 2>&1 | grep "^{" > gpurun_out/r3x_probe28.log
cut -c1-520 gpurun_out/r3x_probe28.log
