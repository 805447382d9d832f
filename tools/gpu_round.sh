#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprofv3 kernel stats of the same bench command.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
( time timeout 900 python bench.py --steps 1 --warmup 1 ) > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/bench.log
ROOT=$PWD
cd /tmp
( time timeout 1200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > $ROOT/gpurun_out/bench_prof.log 2>&1
cd $ROOT
tail -3 gpurun_out/bench_prof.log
find gpurun_out/prof -name "*stats*" | head
# keep only the small stats files (the raw trace is large)
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
DB=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/kernel_stats.csv 2>gpurun_out/kernel_stats.err
head -20 gpurun_out/kernel_stats.csv
find gpurun_out/prof -size +30M -delete
du -sh gpurun_out
