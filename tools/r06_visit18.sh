#!/bin/bash
# round 6, visit 18: decode attention at 32 requests -- chunks per workgroup (the rule caps at 8: 640 workgroups of 8 x 128 keys at 4 936-token contexts)
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/v18_attn_decode_cpw.log
for c in auto 3 4 5 6 7 8 10 13; do
  if [ $c = auto ]; then ( timeout 300 python tools/attn_decode_bench.py 32 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1
  else ( BAGEL_DEC_CPW=$c timeout 300 python tools/attn_decode_bench.py 32 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1; fi
done
for c in auto 5; do
  if [ $c = auto ]; then ( timeout 300 python tools/attn_decode_bench.py 16 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1
  else ( BAGEL_DEC_CPW=$c timeout 300 python tools/attn_decode_bench.py 16 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1; fi
done
grep -v amdgpu.ids gpurun_out/v18_attn_decode_cpw.log
