#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1
tail -8 gpurun_out/pytest_gpu.log
( time timeout 600 python tools/train_forward_probe.py ) > gpurun_out/train_forward.log 2>&1
tail -5 gpurun_out/train_forward.log
