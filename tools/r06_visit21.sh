#!/bin/bash
# round 6, visit 21: the driver's pytest command + smoke on the FINAL commit (after the decode-attention cap and the 20-request parity case)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/v21_pytest_single.log 2>&1
echo "pytest rc=$?" > gpurun_out/v21_rc.txt
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/v21_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/v21_rc.txt
cat gpurun_out/v21_rc.txt; tail -4 gpurun_out/v21_pytest_single.log; tail -2 gpurun_out/v21_smoke.log
