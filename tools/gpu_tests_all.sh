#!/bin/bash
# The whole GPU suite, one pytest process per file (a fault in one file cannot take the others down), summary lines only.  Log -> gpurun_out/pytest_gpu.log
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pytest_gpu.log
for f in tests/test_*_gpu.py; do
  echo "=== $f" >> gpurun_out/pytest_gpu.log
  ( timeout 1200 python -m pytest $f -m gpu -q --timeout 900 ) >> gpurun_out/pytest_gpu.log 2>&1
  echo "$f: $(grep -E 'passed|failed|error|Abort|Timeout' gpurun_out/pytest_gpu.log | tail -1)"
done
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20
