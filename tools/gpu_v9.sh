#!/bin/bash
# Round-4 visit 9: bf16 GroupNorm with 1024 slices (tests + timing); gemv_mb weight loads plain vs non-temporal (interleaved processes).
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_und_shapes_gpu.py tests/test_inferencer_gpu.py tests/test_gemv_mb_gpu.py -m gpu -q --timeout 600 -k "vae or bf16 or inferencer or gemv_mb" ) > gpurun_out/v9_pytest.log 2>&1; tail -3 gpurun_out/v9_pytest.log | cut -c1-300
( timeout 600 python tools/vae_bench.py ) > gpurun_out/v9_vae_bench.log 2>&1; grep -v amdgpu gpurun_out/v9_vae_bench.log | tail -2
for rep in 1 2; do
  for nt in 0 1; do
    ( BAGEL_MB_NT=$nt timeout 600 python tools/gemv_mb_bench.py 2 16 ) > gpurun_out/v9_mb_nt${nt}_$rep.log 2>&1
    echo "--- BAGEL_MB_NT=$nt rep $rep"; grep -v amdgpu gpurun_out/v9_mb_nt${nt}_$rep.log | cut -c1-400 | tail -8
  done
done
if [ -f tests/golden/wide7b_traj49.pt ]; then
  ( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 800 -k "49_step" -s ) > gpurun_out/v9_pytest_traj.log 2>&1; grep -E "drift|^ +[0-9]+ \||passed|failed|Error" gpurun_out/v9_pytest_traj.log | cut -c1-300
fi
