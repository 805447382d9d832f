#!/bin/bash
# Round-4 visit 6: the bf16-autocast VAE (tests, 1024^2 timing), the 49-step trajectory test if its fixture is there, gemv_mb vs the lane-FMA gemv.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_und_shapes_gpu.py tests/test_inferencer_gpu.py tests/test_image_io_gpu.py -m gpu -q --timeout 600 -s ) > gpurun_out/v6_pytest_vae.log 2>&1; grep -E "bf16-autocast|mean .diff|passed|failed|Error|assert" gpurun_out/v6_pytest_vae.log | cut -c1-400 | tail -14
( timeout 600 python tools/vae_bench.py ) > gpurun_out/v6_vae_bench.log 2>&1; grep -v amdgpu gpurun_out/v6_vae_bench.log | tail -3
if [ -f tests/golden/wide7b_traj49.pt ]; then
  ( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 800 -k "49_step" -s ) > gpurun_out/v6_pytest_traj.log 2>&1; grep -E "drift|^ +[0-9]+ \||passed|failed|Error" gpurun_out/v6_pytest_traj.log | cut -c1-300
fi
( timeout 600 python tools/gemv_mb_bench.py 2 4 ) > gpurun_out/v6_gemv_mb_bench.log 2>&1; grep -v amdgpu gpurun_out/v6_gemv_mb_bench.log
find gpurun_out -size +5M -delete
