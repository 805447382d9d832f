#!/bin/bash
# Round-4 visit 2: SGPR-base DMA as variant 5 (bit identity, benchmark-shape tests, in-process A/B vs variant 4, M0 handling builds),
# decode attention with several chunks per workgroup (tests, B = 16 / 2 / 1 whole-step timing over the chunk counts).
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -k "gemm" ) > gpurun_out/v2_pytest_ops.log 2>&1; tail -3 gpurun_out/v2_pytest_ops.log
( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 600 -k "gemm_persistent" ) > gpurun_out/v2_pytest_wide.log 2>&1; tail -3 gpurun_out/v2_pytest_wide.log
( timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q --timeout 600 ) > gpurun_out/v2_pytest_decode.log 2>&1; tail -3 gpurun_out/v2_pytest_decode.log
: > gpurun_out/v2_gemm_ab.log
for lib in "" m0g1 m0g2 "" m0g1 m0g2; do
  if [ -z "$lib" ]; then L=$ROOT/bagel_amd/libbagel_hip.so; else L=$ROOT/bagel_amd/libbagel_hip_$lib.so; fi
  ( BAGEL_HIP_LIB=$L timeout 300 python tools/gemm_ab.py 2 ) >> gpurun_out/v2_gemm_ab.log 2>&1
done
grep -v amdgpu.ids gpurun_out/v2_gemm_ab.log
for cpw in 0 1 2 4 8; do
  ( BAGEL_DEC_CPW=$cpw timeout 900 python bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode --und-batch 16 --und-new-tokens 160 ) > gpurun_out/v2_und_b16_cpw$cpw.log 2>&1
  grep "^{" gpurun_out/v2_und_b16_cpw$cpw.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); u=d.get('understanding',d); print('B=16 cpw=$cpw', {k:u.get(k) for k in ('value','decode_ms_per_step')})"
done
( timeout 600 python bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode --und-batch 2 --und-new-tokens 160 ) > gpurun_out/v2_und_b2.log 2>&1
grep "^{" gpurun_out/v2_und_b2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); u=d.get('understanding',d); print('B=2', {k:u.get(k) for k in ('value','decode_ms_per_step')})"
( timeout 600 python bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode ) > gpurun_out/v2_und_b1.log 2>&1
grep "^{" gpurun_out/v2_und_b1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); u=d.get('understanding',d); print('B=1', {k:u.get(k) for k in ('value','decode_ms_per_step','prefill_ms')})"
find gpurun_out -size +5M -delete
