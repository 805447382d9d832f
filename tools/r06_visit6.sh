#!/bin/bash
# round 6, visit 6: rocprofv3 kernel stats of one bench step (without the B = 1 memory leg) + the PMC passes on the final kernel sources
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 0 --no-memory-leg --no-cpu-baseline --no-understanding --no-taylorseer --no-edit --no-fp8 --no-train-forward ) > $ROOT/gpurun_out/bench_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/bench_kernel_stats.csv 2>gpurun_out/kernel_stats.err
head -8 gpurun_out/bench_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof
timeout 1500 bash tools/gpu_pmc.sh > gpurun_out/v6_pmc.log 2>&1
echo "pmc rc=$?"
find gpurun_out -size +5M -delete
