#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_wide_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 600 -x -k "gemm_leftover or gemm_persistent" -s ) 2>&1 | grep -v "^$" | tail -25
( timeout 300 python tools/gemm_splitk_probe.py ) 2>&1 | grep -v amdgpu
