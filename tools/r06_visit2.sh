#!/bin/bash
# round 6, visit 2: split-ring attention A/B + parity, bench with paused workers + two edit requests per GPU, PMC pass
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/attn2_probe.py --iters 20 --workers 512 > gpurun_out/v2_attn_split.log 2>&1
echo "probe rc=$?" > gpurun_out/v2_rc.txt
timeout 1200 python -m pytest tests/test_attn2_gpu.py -x -q > gpurun_out/v2_attn2_tests.log 2>&1
echo "attn2 tests rc=$?" >> gpurun_out/v2_rc.txt
( time timeout 1700 python bench.py ) > gpurun_out/v2_bench.log 2> gpurun_out/v2_bench.err
echo "bench rc=$?" >> gpurun_out/v2_rc.txt
timeout 1500 bash tools/gpu_pmc.sh > gpurun_out/v2_pmc.log 2>&1
echo "pmc rc=$?" >> gpurun_out/v2_rc.txt
cat gpurun_out/v2_rc.txt
