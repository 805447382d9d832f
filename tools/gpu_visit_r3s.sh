#!/bin/bash
# round 3, visit s: forward-produced lse + vectorised statistics reads in the attention reverse: parity, probe, 4-layer step
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_backward_gpu.py tests/test_train_gpu.py -q -k "attention or gradients or block_mask or training_forward" 2>&1 | tail -8 > gpurun_out/r3s_pytest.log
tail -8 gpurun_out/r3s_pytest.log
python tools/attn_bwd_probe.py 2>&1 | tail -1
PROBE_LAYERS=4 PROBE_ITERS=2 timeout 600 python tools/train_step_probe.py 2>&1 | grep "^{" | cut -c1-420
