#!/bin/bash
# round 6, visit 20: python bench.py (no flags) on the FINAL commit -- a second sample of the final sources on another box
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1790 python bench.py ) > gpurun_out/v20_bench_default.log 2> gpurun_out/v20_bench_default.err
echo "bench rc=$?" > gpurun_out/v20_rc.txt
cat gpurun_out/v20_rc.txt; tail -4 gpurun_out/v20_bench_default.err
