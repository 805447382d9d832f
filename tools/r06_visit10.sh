#!/bin/bash
# round 6, visit 10: (a) gemv_mb with two blocks of request rows (17..32 rows per weight pass): parity, stand-alone timing, the batched decode at 16 / 32 requests;
# (b) the ViT's epilogues (bias + residual, bias + GELU-tanh) on the persistent GEMM: parity, bit identity with the one-tile kernel, per-shape timing, prefill A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gemv_mb_gpu.py tests/test_decode_gpu.py tests/test_ops_gpu.py -x -q -k "gemv_mb or decode or gemm" ) > gpurun_out/v10_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/v10_rc.txt
( timeout 600 python tools/gemv_mb_bench.py 16 32 ) > gpurun_out/v10_gemv_mb_bench.log 2>&1
echo "bench_mb rc=$?" >> gpurun_out/v10_rc.txt
( timeout 600 python tools/gemm_prefill_shapes.py ) > gpurun_out/v10_vit_shapes.log 2>&1
echo "shapes rc=$?" >> gpurun_out/v10_rc.txt
for k in 1 0 1 0; do
  ( BAGEL_GEMM_VIT_EPI=$k timeout 900 python bench.py --gpus 1 --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode ) > gpurun_out/v10_und_vitepi$k.log 2>> gpurun_out/v10_und.err
  echo "und vitepi=$k rc=$?" >> gpurun_out/v10_rc.txt
  grep -o '"prefill_ms": {[^}]*}' gpurun_out/v10_und_vitepi$k.log | tail -1 >> gpurun_out/v10_prefill_ab.log
done
( time timeout 900 python bench.py --gpus 1 --only-understanding --no-cpu-baseline ) > gpurun_out/v10_und.log 2>> gpurun_out/v10_und.err
echo "und rc=$?" >> gpurun_out/v10_rc.txt
cat gpurun_out/v10_rc.txt; tail -5 gpurun_out/v10_tests.log; tail -8 gpurun_out/v10_gemv_mb_bench.log; cat gpurun_out/v10_vit_shapes.log | cut -c1-260; cat gpurun_out/v10_prefill_ab.log
grep -o '"batched_decode[_0-9]*": {[^}]*' gpurun_out/v10_und.log | cut -c1-300
