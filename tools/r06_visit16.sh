#!/bin/bash
# round 6, visit 16: python bench.py (no flags) after the memory hand-back in front of the understanding child
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1790 python bench.py ) > gpurun_out/v16_bench_default.log 2> gpurun_out/v16_bench_default.err
echo "bench rc=$?" > gpurun_out/v16_rc.txt
cat gpurun_out/v16_rc.txt; tail -4 gpurun_out/v16_bench_default.err
