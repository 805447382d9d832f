#!/bin/bash
# Round-4 visit 11: the 49-step trajectory at 7B width against the unmodified reference (tests/golden/wide7b_traj49.pt)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 800 -k "49_step" -s ) > gpurun_out/v11_pytest_traj.log 2>&1; grep -E "drift|^ +[0-9]+ \||passed|failed|Error" gpurun_out/v11_pytest_traj.log | cut -c1-400
