#!/bin/bash
# GPU visit 8: batched decode scaling (understanding leg at batch 1/2/4/8 per GPU).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for ub in 1 2 4 8; do
  echo "== und-batch $ub" >> gpurun_out/und_batch.log
  ( timeout 300 python bench.py --only-understanding --und-batch $ub 2>&1 | grep '^{' | python -c "import json,sys; u=json.loads(sys.stdin.read())['understanding']; print(json.dumps({k:u.get(k) for k in ('value','decode_ms_per_step','decode_ms_per_token','prefill_ms','hip_graph','error','trace')})); print(u.get('roofline'))" ) >> gpurun_out/und_batch.log 2>&1
done
cat gpurun_out/und_batch.log
