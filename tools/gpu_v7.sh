#!/bin/bash
# Round-4 visit 7: kernel stats of the bf16 / fp32 VAE at 1024^2; the reworked inferencer test; the 49-step test if the fixture exists.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( timeout 600 python -m pytest tests/test_inferencer_gpu.py -m gpu -q --timeout 600 -s -k "bf16" ) > gpurun_out/v7_pytest_inf.log 2>&1; grep -E "mean .diff|passed|failed|Error" gpurun_out/v7_pytest_inf.log | cut -c1-300
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_vae -o vae -- python $ROOT/tools/vae_bench.py ) > $ROOT/gpurun_out/v7_vae_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_vae -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/v7_vae_kernel_stats.csv 2>gpurun_out/v7_err.log
head -24 gpurun_out/v7_vae_kernel_stats.csv | cut -c1-130
rm -rf gpurun_out/prof_vae
if [ -f tests/golden/wide7b_traj49.pt ]; then
  ( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 800 -k "49_step" -s ) > gpurun_out/v7_pytest_traj.log 2>&1; grep -E "drift|^ +[0-9]+ \||passed|failed|Error" gpurun_out/v7_pytest_traj.log | cut -c1-300
fi
find gpurun_out -size +5M -delete
