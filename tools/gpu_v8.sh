#!/bin/bash
# Round-4 visit 8: bf16 VAE after the GroupNorm rework (tests + timing + kernel stats).
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( timeout 900 python -m pytest tests/test_und_shapes_gpu.py tests/test_inferencer_gpu.py -m gpu -q --timeout 600 -s -k "vae or bf16 or inferencer" ) > gpurun_out/v8_pytest_vae.log 2>&1; grep -E "bf16-autocast|mean .diff|passed|failed|Error" gpurun_out/v8_pytest_vae.log | cut -c1-300 | tail -10
( timeout 600 python tools/vae_bench.py ) > gpurun_out/v8_vae_bench.log 2>&1; grep -v amdgpu gpurun_out/v8_vae_bench.log | tail -2
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_vae -o vae -- python $ROOT/tools/vae_bench.py ) > $ROOT/gpurun_out/v8_vae_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_vae -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/v8_vae_kernel_stats.csv 2>gpurun_out/v8_err.log
head -16 gpurun_out/v8_vae_kernel_stats.csv | cut -c1-130
rm -rf gpurun_out/prof_vae
if [ -f tests/golden/wide7b_traj49.pt ]; then
  ( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 800 -k "49_step" -s ) > gpurun_out/v8_pytest_traj.log 2>&1; grep -E "drift|^ +[0-9]+ \||passed|failed|Error" gpurun_out/v8_pytest_traj.log | cut -c1-300
fi
find gpurun_out -size +5M -delete
