#!/bin/bash
# Round-4 visit 13: 49-step trajectory test on the refined fixture; gemv_mb workgroup-count sensitivity
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 800 -k "49_step" -s ) > gpurun_out/v13_pytest_traj.log 2>&1; grep -E "drift|^ +[0-9]+ \||passed|failed|Error" gpurun_out/v13_pytest_traj.log | cut -c1-400
( timeout 300 python tools/traj_probe.py ) 2>&1 | grep -E "^rep" | cut -c1-600 > gpurun_out/v13_traj_probe.log; cat gpurun_out/v13_traj_probe.log
for wgs in 256 148 192 224 296 512; do
  echo "--- BAGEL_MB_WGS=$wgs"
  ( BAGEL_MB_WGS=$wgs timeout 600 python tools/gemv_mb_bench.py 2 16 ) 2>&1 | grep -v amdgpu | cut -c1-400 | tail -9
done > gpurun_out/v13_mb_wgs.log 2>&1
cat gpurun_out/v13_mb_wgs.log
