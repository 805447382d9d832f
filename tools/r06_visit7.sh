#!/bin/bash
# round 6, visit 7: the driver's own commands on the final sources -- ONE pytest process over tests/ -m gpu, then python bench.py --gpus 1 --steps 20 --warmup 5
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v7_pytest_single.log 2>&1
echo "pytest rc=$?" > gpurun_out/v7_rc.txt
( time timeout 1790 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/v7_bench_driverlike.log 2> gpurun_out/v7_bench_driverlike.err
echo "bench rc=$?" >> gpurun_out/v7_rc.txt
cat gpurun_out/v7_rc.txt; tail -4 gpurun_out/v7_pytest_single.log; tail -4 gpurun_out/v7_bench_driverlike.err
