#!/bin/bash
# GPU visit 5: gemv v4 parity + decode variants + per-shape timing.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( time timeout 600 python -m pytest tests/test_decode_gpu.py -q --timeout 300 -x ) > gpurun_out/pytest_decode.log 2>&1
tail -15 gpurun_out/pytest_decode.log
for v in "A=1" "BAGEL_GEMV_SPLITK=0" "BAGEL_DEC_CH=64" "BAGEL_GEMV_WGS=2048" "BAGEL_GEMV_WGS=1024"; do
  echo "== $v" >> gpurun_out/variants.log
  ( env $v timeout 300 python bench.py --only-understanding 2>&1 | grep '^{' | python -c "import json,sys; u=json.loads(sys.stdin.read())['understanding']; print(u.get('value'), u.get('decode_ms_per_token'), u.get('error'))" ) >> gpurun_out/variants.log 2>&1
done
cat gpurun_out/variants.log
cd /tmp
( time timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_und -o und -- python $ROOT/bench.py --only-understanding --und-new-tokens 64 ) > $ROOT/gpurun_out/prof_und.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_und -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/und_kernel_stats.csv 2>gpurun_out/und_kernel_stats.err
[ -n "$DB" ] && python tools/rocprof_by_grid.py $DB gemv > gpurun_out/und_gemv_by_grid.csv 2>gpurun_out/und_gemv_by_grid.err
rm -rf gpurun_out/prof_und
find gpurun_out -size +5M -delete
