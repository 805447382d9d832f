#!/bin/bash
# round 3, visit q: attention reverse with prefetch + mask-free tiles: parity, then kernel stats of a 4-layer training step
mkdir -p gpurun_out
ROOT=$PWD
timeout 900 python -m pytest tests/test_train_backward_gpu.py -q -k "attention or gradients" 2>&1 | tail -8 > gpurun_out/r3q_pytest.log
tail -8 gpurun_out/r3q_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ts
PROBE_LAYERS=4 PROBE_ITERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python $ROOT/tools/train_step_probe.py > $ROOT/gpurun_out/r3q_prof_run.log 2>&1
cd $ROOT
grep "^{" gpurun_out/r3q_prof_run.log | cut -c1-600
DB=$(find /tmp/prof_ts -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/r3q_train_step_kernel_stats.csv 2>gpurun_out/r3q_err.log
head -12 gpurun_out/r3q_train_step_kernel_stats.csv | cut -c1-150
