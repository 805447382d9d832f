#!/bin/bash
# Round-4 visit 1: new batched-decode projection (tests, per-shape timing, whole step), GEMM A/B builds (SGPR-base DMA, epilogue ablations),
# idle-gap analysis of one benchmark step.  Logs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( timeout 600 python -m pytest tests/test_gemv_mb_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/v1_pytest_mb.log 2>&1; tail -3 gpurun_out/v1_pytest_mb.log
( timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --timeout 600 ) > gpurun_out/v1_pytest_decode.log 2>&1; tail -3 gpurun_out/v1_pytest_decode.log
( timeout 600 python tools/gemv_mb_bench.py 2 16 ) > gpurun_out/v1_gemv_mb_bench.log 2>&1; cat gpurun_out/v1_gemv_mb_bench.log
: > gpurun_out/v1_gemm_ab.log
for lib in "" saddr abl1 abl2 abl3 "" saddr; do
  if [ -z "$lib" ]; then L=$ROOT/bagel_amd/libbagel_hip.so; else L=$ROOT/bagel_amd/libbagel_hip_$lib.so; fi
  ( BAGEL_HIP_LIB=$L timeout 300 python tools/gemm_ab.py 2 ) >> gpurun_out/v1_gemm_ab.log 2>&1
done
cat gpurun_out/v1_gemm_ab.log
for mb in 1 0; do
  ( BAGEL_GEMV_MB=$mb timeout 900 python bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode --und-batch 16 --und-new-tokens 96 ) > gpurun_out/v1_und_b16_mb$mb.log 2>&1
  grep "^{" gpurun_out/v1_und_b16_mb$mb.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); u=d.get('understanding',d); print('MB=$mb', {k:u.get(k) for k in ('value','decode_ms_per_step','prefill_ms')})"
done
( BAGEL_GEMV_MB=1 timeout 600 python bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode --und-batch 2 --und-new-tokens 96 ) > gpurun_out/v1_und_b2.log 2>&1
grep "^{" gpurun_out/v1_und_b2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); u=d.get('understanding',d); print('B=2', {k:u.get(k) for k in ('value','decode_ms_per_step')})"
cd /tmp
( BAGEL_GEMV_MB=1 timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_b16 -o und -- python $ROOT/bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode --und-batch 16 --und-new-tokens 48 ) > $ROOT/gpurun_out/v1_und_b16_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_b16 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/v1_und_b16_kernel_stats.csv 2>gpurun_out/v1_err.log
head -16 gpurun_out/v1_und_b16_kernel_stats.csv | cut -c1-120
rm -rf gpurun_out/prof_b16
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_t2i -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-understanding --no-taylorseer --no-edit --no-fp8 --no-train-forward ) > $ROOT/gpurun_out/v1_bench_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_t2i -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python tools/rocprof_gaps.py $DB 20 > gpurun_out/v1_bench_gaps.txt 2>>gpurun_out/v1_err.log
  python tools/rocprof_summary.py $DB > gpurun_out/v1_bench_kernel_stats.csv 2>>gpurun_out/v1_err.log
fi
cat gpurun_out/v1_bench_gaps.txt | head -20
grep "^{" gpurun_out/v1_bench_prof.log | cut -c1-300
rm -rf gpurun_out/prof_t2i
find gpurun_out -size +5M -delete
du -sh gpurun_out
