#!/bin/bash
# round 3, visit o: kernel stats of a 4-layer training step (forward with tape + backward)
mkdir -p gpurun_out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ts
PROBE_LAYERS=4 PROBE_ITERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python $ROOT/tools/train_step_probe.py > $ROOT/gpurun_out/r3o_prof_run.log 2>&1
cd $ROOT
DB=$(find /tmp/prof_ts -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/r3o_train_step_kernel_stats.csv 2>gpurun_out/r3o_err.log
head -40 gpurun_out/r3o_train_step_kernel_stats.csv | cut -c1-180
