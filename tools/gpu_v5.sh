#!/bin/bash
# Round-4 visit 5: attention epilogue with 16-byte stores (tests + A/B), buffer_load-lds DMA experiment on the GEMM, decode phase probe over cpw.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( timeout 900 python -m pytest tests/test_attn2_gpu.py tests/test_ops_gpu.py tests/test_wide_gpu.py -m gpu -q -x --timeout 600 -k "attn or attention" ) > gpurun_out/v5_pytest_attn.log 2>&1; tail -2 gpurun_out/v5_pytest_attn.log
: > gpurun_out/v5_attn_ab.log
for lib in "" a2w0 "" a2w0; do
  if [ -z "$lib" ]; then L=$ROOT/bagel_amd/libbagel_hip.so; else L=$ROOT/bagel_amd/libbagel_hip_$lib.so; fi
  echo "== $(basename $L)" >> gpurun_out/v5_attn_ab.log
  ( BAGEL_HIP_LIB=$L timeout 300 python tools/attn2_probe.py --iters 20 ) >> gpurun_out/v5_attn_ab.log 2>&1
done
grep -v amdgpu gpurun_out/v5_attn_ab.log | cut -c1-150
: > gpurun_out/v5_gemm_ab.log
for lib in "" buf "" buf; do
  if [ -z "$lib" ]; then L=$ROOT/bagel_amd/libbagel_hip.so; else L=$ROOT/bagel_amd/libbagel_hip_$lib.so; fi
  ( BAGEL_HIP_LIB=$L timeout 300 python tools/gemm_ab.py 2 ) >> gpurun_out/v5_gemm_ab.log 2>&1
done
grep -v amdgpu gpurun_out/v5_gemm_ab.log
( BAGEL_HIP_LIB=$ROOT/bagel_amd/libbagel_hip_buf.so timeout 600 python tools/gemm_persist_check.py ) > gpurun_out/v5_buf_check.log 2>&1; tail -1 gpurun_out/v5_buf_check.log
for cpw in 0 2 5 0 2 5; do
  ( BAGEL_DEC_CPW=$cpw timeout 600 python tools/decode_phase_probe.py 16 160 ) > gpurun_out/v5_phase_cpw$cpw.log 2>&1; echo "cpw=$cpw $(grep '^rep 1' gpurun_out/v5_phase_cpw$cpw.log | cut -c1-230)"
done
find gpurun_out -size +5M -delete
