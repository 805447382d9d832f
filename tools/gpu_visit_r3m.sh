#!/bin/bash
# round 3, visit m: first GPU run of the training backward (kernels, attention reverse, whole-step gradients) + a timing probe
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_backward_gpu.py -q -s 2>&1 | tail -60 > gpurun_out/r3m_pytest.log
tail -25 gpurun_out/r3m_pytest.log
PROBE_LAYERS=4 timeout 600 python tools/train_step_probe.py > gpurun_out/r3m_probe4.log 2>&1
tail -3 gpurun_out/r3m_probe4.log
