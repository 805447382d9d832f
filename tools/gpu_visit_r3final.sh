#!/bin/bash
# round 3, closing visit: the whole GPU suite once more on the final sources, smoke, kernel stats of a full-depth training step
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
: > gpurun_out/pytest_gpu.log
for f in tests/test_*_gpu.py; do
  echo "=== $f" >> gpurun_out/pytest_gpu.log
  ( timeout 900 python -m pytest $f -m gpu -q --timeout 600 ) >> gpurun_out/pytest_gpu.log 2>&1
  echo "$f: $(grep -E 'passed|failed|error|Abort' gpurun_out/pytest_gpu.log | tail -1)"
done
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
cd /tmp
rm -rf /tmp/prof_ts
PROBE_ITERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python $ROOT/tools/train_step_probe.py > $ROOT/gpurun_out/final_train_prof_run.log 2>&1
cd $ROOT
grep "^{" gpurun_out/final_train_prof_run.log | cut -c100-400
DB=$(find /tmp/prof_ts -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/train_step_kernel_stats.csv 2>/dev/null
head -24 gpurun_out/train_step_kernel_stats.csv | cut -c1-130
