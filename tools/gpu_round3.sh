#!/bin/bash
# GPU visit 3: decode optimisations + TaylorSeer parity, full bench line, decode kernel stats.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( time timeout 600 python -m pytest tests/test_decode_gpu.py -q --timeout 300 ) > gpurun_out/pytest_decode.log 2>&1
tail -40 gpurun_out/pytest_decode.log
( time timeout 900 python bench.py --steps 1 --warmup 1 ) > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/bench.log
cd /tmp
( time timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_und -o und -- python $ROOT/bench.py --only-understanding --und-new-tokens 64 ) > $ROOT/gpurun_out/prof_und.log 2>&1
cd $ROOT
tail -3 gpurun_out/prof_und.log
DB=$(find gpurun_out/prof_und -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/und_kernel_stats.csv 2>gpurun_out/und_kernel_stats.err
head -16 gpurun_out/und_kernel_stats.csv
rm -rf gpurun_out/prof_und
find gpurun_out -size +5M -delete
du -sh gpurun_out
