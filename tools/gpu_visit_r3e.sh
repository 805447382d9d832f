#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_fp8_gpu.py tests/test_full_depth_gpu.py -m gpu -q --timeout 600 -x ) 2>&1 | tail -40
echo "=== inferencer with debug sync"
( BAGEL_DEBUG_SYNC=1 timeout 600 python -m pytest tests/test_inferencer_gpu.py -m gpu -q --timeout 600 -x -k think ) 2>&1 | grep -v "^  File\|pluggy\|_pytest" | tail -30
echo "=== inferencer, tile kernel"
( BAGEL_ATTN_PLANNED=0 timeout 600 python -m pytest tests/test_inferencer_gpu.py -m gpu -q --timeout 600 -x ) 2>&1 | tail -3
