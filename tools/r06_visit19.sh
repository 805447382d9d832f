#!/bin/bash
# round 6, visit 19: decode attention auto-cpw capped at 5: decode tests, batched decode 16 / 32, PMC passes (decode.hip changed)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_gemv_mb_gpu.py -x -q ) > gpurun_out/v19_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/v19_rc.txt
( timeout 900 python bench.py --gpus 1 --only-understanding --no-cpu-baseline --no-int8 ) > gpurun_out/v19_und.log 2>> gpurun_out/v19_und.err
echo "und rc=$?" >> gpurun_out/v19_rc.txt
timeout 1500 bash tools/gpu_pmc.sh > gpurun_out/v19_pmc.log 2>&1
echo "pmc rc=$?" >> gpurun_out/v19_rc.txt
cat gpurun_out/v19_rc.txt; tail -2 gpurun_out/v19_tests.log
grep -o '"batched_decode[_0-9]*": {[^}]*' gpurun_out/v19_und.log | cut -c1-200
grep -o '"value": [0-9.]*, "unit": "tokens/s", "per_gpu' gpurun_out/v19_und.log
