#!/bin/bash
# Round-4 visit 10: gemv_mb with fragment-major (tiled) weight addressing, timing only (BAGEL_MB_TILED_TIMING=1 reads row-major data as if tiled)
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
  for tl in 0 1; do
    ( BAGEL_MB_TILED_TIMING=$tl timeout 600 python tools/gemv_mb_bench.py 2 16 ) > gpurun_out/v10_mb_tiled${tl}_$rep.log 2>&1
    echo "--- BAGEL_MB_TILED_TIMING=$tl rep $rep"; grep -v amdgpu gpurun_out/v10_mb_tiled${tl}_$rep.log | cut -c1-400 | tail -9
  done
done
( timeout 600 python -m pytest tests/test_decode_gpu.py -m gpu -q -x --timeout 600 ) > gpurun_out/v10_pytest_decode.log 2>&1; tail -2 gpurun_out/v10_pytest_decode.log
( timeout 600 python tools/decode_phase_probe.py 16 ) > gpurun_out/v10_phase_b16.log 2>&1; grep "^rep" gpurun_out/v10_phase_b16.log | cut -c1-400
if [ -f tests/golden/wide7b_traj49.pt ]; then
  ( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 800 -k "49_step" -s ) > gpurun_out/v10_pytest_traj.log 2>&1; grep -E "drift|^ +[0-9]+ \||passed|failed|Error" gpurun_out/v10_pytest_traj.log | cut -c1-300
fi
