#!/bin/bash
# round 6, last visit: the driver's pytest command and `python bench.py` (no flags) on the FINAL commit
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v12_pytest_single.log 2>&1
echo "pytest rc=$?" > gpurun_out/v12_rc.txt
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/v12_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/v12_rc.txt
( time timeout 1790 python bench.py ) > gpurun_out/v12_bench_default.log 2> gpurun_out/v12_bench_default.err
echo "bench rc=$?" >> gpurun_out/v12_rc.txt
cat gpurun_out/v12_rc.txt; tail -3 gpurun_out/v12_pytest_single.log; tail -2 gpurun_out/v12_smoke.log; tail -4 gpurun_out/v12_bench_default.err
