#!/bin/bash
# One GPU visit for the record: the full parity suite, smoke(), the full bench line (every leg), rocprofv3 kernel stats of the same
# bench command without the extra legs.  PMC passes: tools/gpu_pmc.sh.   Logs -> gpurun_out/, to be copied into profiles/r0N_*.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
if [ -z "$SKIP_PYTEST" ]; then
  ( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1
  grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -3
  ( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
fi
( time timeout 1500 python bench.py --steps ${STEPS:-2} --warmup 1 ) > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-600
if [ -z "$SKIP_PROF" ]; then
  cd /tmp
  ( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-understanding --no-taylorseer --no-edit --no-fp8 ) > $ROOT/gpurun_out/bench_prof.log 2>&1
  cd $ROOT
  DB=$(find gpurun_out/prof -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/bench_kernel_stats.csv 2>gpurun_out/kernel_stats.err
  head -12 gpurun_out/bench_kernel_stats.csv
  rm -rf gpurun_out/prof
fi
find gpurun_out -size +5M -delete
du -sh gpurun_out
