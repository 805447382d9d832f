#!/bin/bash
# round 3, visit n: training step at full depth + kernel stats of a 4-layer step
mkdir -p gpurun_out
ROOT=$PWD
PROBE_ITERS=2 timeout 900 python tools/train_step_probe.py > gpurun_out/r3n_probe28.log 2>&1
tail -1 gpurun_out/r3n_probe28.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ts
PROBE_LAYERS=4 PROBE_ITERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python $ROOT/tools/train_step_probe.py > $ROOT/gpurun_out/r3n_prof_run.log 2>&1
F=$(find /tmp/prof_ts -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" $ROOT/gpurun_out/r3n_train_step_kernel_stats.csv
head -30 $ROOT/gpurun_out/r3n_train_step_kernel_stats.csv | cut -c1-200
