#!/bin/bash
# round 3, visit u: launch-ordered GEMM durations of a 2-layer training step
mkdir -p gpurun_out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ts
PROBE_LAYERS=2 PROBE_ITERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python $ROOT/tools/train_step_probe.py > $ROOT/gpurun_out/r3v_prof_run.log 2>&1
cd $ROOT
DB=$(find /tmp/prof_ts -name "*.db" | head -1)
python tools/rocprof_sequence.py $DB "" 130 > gpurun_out/r3v_sequence.log 2>&1
python tools/rocprof_summary.py $DB > gpurun_out/r3v_kernel_stats.csv 2>/dev/null
head -8 gpurun_out/r3v_kernel_stats.csv | cut -c1-120
