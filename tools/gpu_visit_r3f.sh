#!/bin/bash
# the whole GPU suite, one pytest process per file (a device fault then costs one file, not the run)
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/r3f_pytest.log
for f in tests/test_*_gpu.py; do
  echo "=== $f" >> gpurun_out/r3f_pytest.log
  ( timeout 900 python -m pytest $f -m gpu -q --timeout 600 ) >> gpurun_out/r3f_pytest.log 2>&1
  echo "$f: $(grep -E 'passed|failed|error|Abort' gpurun_out/r3f_pytest.log | tail -1)"
done
