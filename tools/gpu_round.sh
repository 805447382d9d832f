#!/bin/bash
# Record visit of a round: the whole GPU suite (one pytest process per file), smoke(), the full bench line (every leg), rocprofv3 kernel stats of
# the same bench command without the extra legs, and of the understanding leg.  PMC passes: tools/gpu_pmc.sh.  Logs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
if [ -z "$SKIP_PYTEST" ]; then
  : > gpurun_out/pytest_gpu.log
  for f in tests/test_*_gpu.py; do
    echo "=== $f" >> gpurun_out/pytest_gpu.log
    ( timeout 900 python -m pytest $f -m gpu -q --timeout 600 ) >> gpurun_out/pytest_gpu.log 2>&1
    echo "$f: $(grep -E 'passed|failed|error|Abort' gpurun_out/pytest_gpu.log | tail -1)"
  done
  ( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
fi
( time timeout 2400 python bench.py --steps ${STEPS:-2} --warmup 1 ) > gpurun_out/bench.log 2>&1
grep "^{" gpurun_out/bench.log | cut -c1-400
if [ -z "$SKIP_PROF" ]; then
  cd /tmp
  ( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-memory-leg --no-cpu-baseline --no-understanding --no-taylorseer --no-edit --no-fp8 --no-train-forward ) > $ROOT/gpurun_out/bench_prof.log 2>&1
  cd $ROOT
  DB=$(find gpurun_out/prof -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/bench_kernel_stats.csv 2>gpurun_out/kernel_stats.err
  head -14 gpurun_out/bench_kernel_stats.csv | cut -c1-150
  rm -rf gpurun_out/prof
  cd /tmp
  ( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof2 -o und -- python $ROOT/bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode ) > $ROOT/gpurun_out/und_prof.log 2>&1
  cd $ROOT
  DB=$(find gpurun_out/prof2 -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/understanding_kernel_stats.csv 2>>gpurun_out/kernel_stats.err
  head -14 gpurun_out/understanding_kernel_stats.csv | cut -c1-150
  rm -rf gpurun_out/prof2
fi
find gpurun_out -size +5M -delete
du -sh gpurun_out
