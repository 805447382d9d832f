#!/bin/bash
# round 6, visit 24: the 7B-width batch-invariance test of the batched decode
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_wide_gpu.py -x -q -s -k "batch_invariant" ) > gpurun_out/v24_tests.log 2>&1
echo "tests rc=$?"; grep -E "wide7b batched|passed|failed|Error|assert" gpurun_out/v24_tests.log | head -20
