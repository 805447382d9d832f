#!/bin/bash
# round 3, visit aa: SigLIP backward: kernels + whole-step gradients (tiny, 7B width, SigLIP width) + step time with the tower trainable
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_train_backward_gpu.py -q -s 2>&1 | grep -v Warn | tail -12 > gpurun_out/r3aa_pytest.log
tail -12 gpurun_out/r3aa_pytest.log
PROBE_VIT=1 PROBE_ITERS=2 timeout 900 python tools/train_step_probe.py 2>&1 | grep "^{" > gpurun_out/r3aa_probe28_vit.log; cut -c1-420 gpurun_out/r3aa_probe28_vit.log
