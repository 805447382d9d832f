#!/bin/bash
# round 3, visit B: the planned persistent attention kernel -- parity tests, then old vs new timing at the benchmark launches; the seam test again
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_attn2_gpu.py tests/test_seam_gpu.py -m gpu -q --timeout 600 -s -x ) > gpurun_out/r3b_pytest.log 2>&1
tail -25 gpurun_out/r3b_pytest.log
( timeout 600 python tools/attn2_probe.py ) > gpurun_out/r3b_probe.log 2>&1
cat gpurun_out/r3b_probe.log
