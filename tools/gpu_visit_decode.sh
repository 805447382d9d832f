#!/bin/bash
# One short GPU visit for the decode path: its parity tests, then the understanding leg under rocprofv3 (kernel stats) for the
# listed environment variants.   usage: bash tools/gpu_visit_decode.sh [VAR=val ...]   (each argument = one extra variant run)
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/decode_visit.log
: > $OUT
( timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_decode.log 2>&1
tail -5 gpurun_out/pytest_decode.log | tee -a $OUT
run() {
  tag=$1; shift
  cd /tmp
  ( env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$tag -o und -- python $ROOT/bench.py --only-understanding --no-int8 --no-cpu-baseline --und-new-tokens 128 ) > $ROOT/gpurun_out/dec_$tag.log 2>&1
  cd $ROOT
  DB=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
  echo "== $tag" >> $OUT
  grep -o '"decode_ms_per_token": [0-9.]*' gpurun_out/dec_$tag.log | head -1 >> $OUT
  grep -o '"hip_graph_error": [^,]*' gpurun_out/dec_$tag.log | head -1 >> $OUT
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB 2>/dev/null | grep "attn_decode\|decode_qkv_post\|gemv_kernel" >> $OUT
  rm -rf gpurun_out/prof_$tag
}
run default A=1
i=0
for v in "$@"; do i=$((i+1)); run var$i $v; done
cat $OUT
