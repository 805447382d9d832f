#!/bin/bash
# GPU visit 2: decode-path parity, understanding bench, its kernel stats, PMC passes on a reduced denoise run.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( time timeout 600 python -m pytest tests/test_decode_gpu.py -q -x --timeout 300 ) > gpurun_out/pytest_decode.log 2>&1
tail -60 gpurun_out/pytest_decode.log
( time timeout 600 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_decode_gpu.py ) > gpurun_out/pytest_gpu.log 2>&1
tail -8 gpurun_out/pytest_gpu.log
( time timeout 600 python bench.py --only-understanding ) > gpurun_out/bench_und.log 2>&1
tail -4 gpurun_out/bench_und.log
cd /tmp
( time timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_und -o und -- python $ROOT/bench.py --only-understanding --und-new-tokens 64 ) > $ROOT/gpurun_out/prof_und.log 2>&1
cd $ROOT
tail -3 gpurun_out/prof_und.log
DB=$(find gpurun_out/prof_und -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/und_kernel_stats.csv 2>gpurun_out/und_kernel_stats.err
head -24 gpurun_out/und_kernel_stats.csv
# PMC passes (own runs, kernel-trace only) on a reduced denoise: counters are per launch, shapes are the full ones
RED="--layers 2 --num-timesteps 3 --no-vae --no-understanding --no-cpu-baseline --warmup 0 --steps 1"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  cd /tmp
  ( time timeout 400 rocprofv3 --kernel-trace --pmc $pass -d $ROOT/gpurun_out/pmc_$tag -o pmc -- python $ROOT/bench.py $RED ) > $ROOT/gpurun_out/pmc_$tag.log 2>&1
  cd $ROOT
  tail -2 gpurun_out/pmc_$tag.log
  DB=$(find gpurun_out/pmc_$tag -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB > gpurun_out/pmc_$tag.txt 2>gpurun_out/pmc_$tag.err
  head -40 gpurun_out/pmc_$tag.txt
  tail -3 gpurun_out/pmc_$tag.err
done
rm -rf gpurun_out/prof_und gpurun_out/pmc_*/
find gpurun_out -size +5M -delete
du -sh gpurun_out
