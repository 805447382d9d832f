#!/bin/bash
# round 6, visit 18: decode attention at 32 requests -- chunks per workgroup (the rule caps at 8: 640 workgroups of 8 x 128 keys at 4 936-token contexts)
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/v18_attn_decode_cpw.log
for c in auto 3 4 5 6 7 8 10 13; do
  if [ $c = auto ]; then ( timeout 300 python tools/attn_decode_bench.py 32 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1
  else ( BAGEL_DEC_CPW=$c timeout 300 python tools/attn_decode_bench.py 32 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1; fi
done
for c in auto 5; do
  if [ $c = auto ]; then ( timeout 300 python tools/attn_decode_bench.py 16 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1
  else ( BAGEL_DEC_CPW=$c timeout 300 python tools/attn_decode_bench.py 16 ) >> gpurun_out/v18_attn_decode_cpw.log 2>&1; fi
done
grep -v amdgpu.ids gpurun_out/v18_attn_decode_cpw.log
# the 20-request generate_text parity case added after the record visit
( timeout 900 python -m pytest tests/test_decode_gpu.py -x -q -k "generate_text_graph_eager" ) > gpurun_out/v18_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/v18_tests.log
# rocprofv3 kernel stats of the batch-1 understanding leg alone (the record visit's profile included the 16 / 32-request legs)
ROOT=$PWD
cd /tmp
( timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof2 -o und -- python $ROOT/bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode ) > $ROOT/gpurun_out/v18_und_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/v18_understanding_kernel_stats.csv 2>gpurun_out/v18_kernel_stats.err
rm -rf gpurun_out/prof2
find gpurun_out -size +5M -delete
head -8 gpurun_out/v18_understanding_kernel_stats.csv | cut -c1-140
