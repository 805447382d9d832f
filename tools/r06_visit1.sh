#!/bin/bash
# round 6, visit 1: the new parity gates + the bench line with the memory object, the oracle workers and edit.parity_at_depth
mkdir -p gpurun_out
export TMPDIR=/tmp
( nproc; free -g | head -2; df -h /dev/shm /tmp | tail -2; lscpu | grep -i "numa\|model name\|socket" ) > gpurun_out/v1_host.txt 2>&1
timeout 1500 python -m pytest tests/test_full_depth_gpu.py -x -q -s > gpurun_out/v1_full_depth.log 2>&1
echo "full_depth rc=$?" >> gpurun_out/v1_host.txt
timeout 900 python -m pytest tests/test_decode_gpu.py -x -q -k "gumbel or naive_cache or sampl" > gpurun_out/v1_decode.log 2>&1
echo "decode rc=$?" >> gpurun_out/v1_host.txt
( time timeout 1700 python bench.py ) > gpurun_out/v1_bench.log 2> gpurun_out/v1_bench.err
echo "bench rc=$?" >> gpurun_out/v1_host.txt
tail -c 3000 gpurun_out/v1_bench.err
