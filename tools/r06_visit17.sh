#!/bin/bash
# round 6, visit 17 (record): the driver's own commands on the final sources -- ONE pytest process over tests/ -m gpu, smoke(), python bench.py --gpus 1 --steps 20 --warmup 5 --
# then rocprofv3 kernel stats of one bench step and of the understanding leg
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v17_pytest_single.log 2>&1
echo "pytest rc=$?" > gpurun_out/v17_rc.txt
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/v17_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/v17_rc.txt
( time timeout 1790 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/v17_bench_driverlike.log 2> gpurun_out/v17_bench_driverlike.err
echo "bench rc=$?" >> gpurun_out/v17_rc.txt
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-memory-leg --no-cpu-baseline --no-understanding --no-taylorseer --no-edit --no-fp8 --no-train-forward ) > $ROOT/gpurun_out/v17_bench_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/v17_bench_kernel_stats.csv 2>gpurun_out/v17_kernel_stats.err
rm -rf gpurun_out/prof
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof2 -o und -- python $ROOT/bench.py --only-understanding --no-cpu-baseline --no-int8 ) > $ROOT/gpurun_out/v17_und_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/v17_understanding_kernel_stats.csv 2>>gpurun_out/v17_kernel_stats.err
rm -rf gpurun_out/prof2
find gpurun_out -size +5M -delete
cat gpurun_out/v17_rc.txt; tail -4 gpurun_out/v17_pytest_single.log; tail -2 gpurun_out/v17_smoke.log; tail -3 gpurun_out/v17_bench_driverlike.err; head -6 gpurun_out/v17_bench_kernel_stats.csv | cut -c1-140; head -8 gpurun_out/v17_understanding_kernel_stats.csv | cut -c1-140
