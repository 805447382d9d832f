#!/bin/bash
# round 6, visit 15: the driver's own commands on the sources of this session -- ONE pytest process over tests/ -m gpu, smoke(), then python bench.py (no flags)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v15_pytest_single.log 2>&1
echo "pytest rc=$?" > gpurun_out/v15_rc.txt
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/v15_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/v15_rc.txt
( time timeout 1790 python bench.py ) > gpurun_out/v15_bench_default.log 2> gpurun_out/v15_bench_default.err
echo "bench rc=$?" >> gpurun_out/v15_rc.txt
cat gpurun_out/v15_rc.txt; tail -4 gpurun_out/v15_pytest_single.log; tail -3 gpurun_out/v15_smoke.log; tail -4 gpurun_out/v15_bench_default.err
