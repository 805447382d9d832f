#!/bin/bash
# One short GPU visit: parity of the register-resident RMSNorm paths, rocprofv3 kernel stats of the decode leg and of one bench step.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py -m gpu -q -k "norm or fp8" --timeout 300 ) > gpurun_out/pytest_norm.log 2>&1
tail -3 gpurun_out/pytest_norm.log
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_und -o und -- python $ROOT/bench.py --only-understanding --no-int8 --no-cpu-baseline ) > $ROOT/gpurun_out/und_prof.log 2>&1
( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-understanding --no-taylorseer --no-edit --no-fp8 ) > $ROOT/gpurun_out/bench_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_und -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/und_kernel_stats.csv 2>gpurun_out/kernel_stats.err
DB=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/bench_kernel_stats.csv 2>>gpurun_out/kernel_stats.err
head -14 gpurun_out/und_kernel_stats.csv
head -12 gpurun_out/bench_kernel_stats.csv
tail -1 gpurun_out/und_prof.log | cut -c1-400
tail -1 gpurun_out/bench_prof.log | cut -c1-300
rm -rf gpurun_out/prof gpurun_out/prof_und
find gpurun_out -size +5M -delete
