#!/bin/bash
# One GPU visit: [full parity suite unless SKIP_PYTEST=1], the full bench line, rocprofv3 kernel stats of the same bench command,
# the GEMM yardsticks.  PMC passes: tools/gpu_pmc.sh.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
if [ -z "$SKIP_PYTEST" ]; then
  ( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1
  tail -6 gpurun_out/pytest_gpu.log
fi
( time timeout 900 python bench.py --steps 1 --warmup 1 ) > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/bench.log
cd /tmp
( time timeout 1200 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-understanding ) > $ROOT/gpurun_out/bench_prof.log 2>&1
cd $ROOT
tail -3 gpurun_out/bench_prof.log
DB=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/kernel_stats.csv 2>gpurun_out/kernel_stats.err
head -24 gpurun_out/kernel_stats.csv
rm -rf gpurun_out/prof
( timeout 200 python tools/gemm_persist_check.py --bench | tail -n 12 ) > gpurun_out/gemm_persistent.log 2>&1
timeout 200 python tools/gemm_vs_library.py > gpurun_out/gemm_vs_library.log 2>&1
cat gpurun_out/gemm_persistent.log gpurun_out/gemm_vs_library.log
find gpurun_out -size +5M -delete
du -sh gpurun_out
