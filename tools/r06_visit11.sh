#!/bin/bash
# round 6, visit 11: gemv_mb two-block form with the weight stream started under the norm's scaling pass; the persistent-kernel ViT epilogues after the check fix
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gemv_mb_gpu.py tests/test_ops_gpu.py -x -q -k "gemv_mb or gemm" ) > gpurun_out/v11_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/v11_rc.txt
( timeout 600 python tools/gemv_mb_bench.py 16 32 ) > gpurun_out/v11_gemv_mb_bench.log 2>&1
echo "bench_mb rc=$?" >> gpurun_out/v11_rc.txt
( time timeout 900 python bench.py --gpus 1 --only-understanding --no-cpu-baseline --no-int8 ) > gpurun_out/v11_und.log 2>> gpurun_out/v11_und.err
echo "und rc=$?" >> gpurun_out/v11_rc.txt
cat gpurun_out/v11_rc.txt; tail -5 gpurun_out/v11_tests.log; tail -8 gpurun_out/v11_gemv_mb_bench.log
grep -o '"batched_decode[_0-9]*": {[^}]*' gpurun_out/v11_und.log | cut -c1-200
grep -o '"prefill_ms": {[^}]*}' gpurun_out/v11_und.log
