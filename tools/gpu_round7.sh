#!/bin/bash
# GPU visit 7: image-io parity, the edit workload (BASELINE configs[4]) bench line.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_decode_gpu.py -q --timeout 300 ) > gpurun_out/pytest_decode.log 2>&1
tail -15 gpurun_out/pytest_decode.log
( time timeout 900 python bench.py --workload edit --steps 1 --warmup 1 --no-understanding --no-cpu-baseline ) > gpurun_out/bench_edit.log 2>&1
tail -3 gpurun_out/bench_edit.log
find gpurun_out -size +5M -delete
