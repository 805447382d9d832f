#!/bin/bash
# round 6, visit 13: gemv_mb reduce tasks dealt over the waves by (output, request block)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gemv_mb_gpu.py tests/test_decode_gpu.py -x -q ) > gpurun_out/v13_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/v13_rc.txt
( timeout 600 python tools/gemv_mb_bench.py 16 32 ) > gpurun_out/v13_gemv_mb_bench.log 2>&1
( time timeout 900 python bench.py --gpus 1 --only-understanding --no-cpu-baseline --no-int8 ) > gpurun_out/v13_und.log 2>> gpurun_out/v13_und.err
echo "und rc=$?" >> gpurun_out/v13_rc.txt
cat gpurun_out/v13_rc.txt; tail -3 gpurun_out/v13_tests.log; tail -8 gpurun_out/v13_gemv_mb_bench.log
grep -o '"batched_decode[_0-9]*": {[^}]*' gpurun_out/v13_und.log | cut -c1-200
