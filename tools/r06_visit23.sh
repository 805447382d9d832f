#!/bin/bash
# round 6, visit 23: bench.py --only-understanding with the HBM roofline objects of the batched legs
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python bench.py --gpus 1 --only-understanding --no-cpu-baseline --no-int8 ) > gpurun_out/v23_und.log 2> gpurun_out/v23_und.err
echo "und rc=$?"
grep -o '"batched_decode[_0-9]*": {[^}]*}[^}]*}' gpurun_out/v23_und.log | cut -c1-420
