#!/bin/bash
# GPU visit: skinny MFMA GEMM parity + batched decode scaling with the wave-target knob.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_model_gpu.py -q --timeout 600 ) > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
for v in "A=1" "BAGEL_SKINNY_WAVES=1500" "BAGEL_SKINNY_WAVES=6000"; do
for ub in 2 8; do
  echo "== $v und-batch $ub" >> gpurun_out/und_batch.log
  ( env $v timeout 300 python bench.py --only-understanding --und-batch $ub 2>&1 | grep '^{' | python -c "import json,sys; u=json.loads(sys.stdin.read())['understanding']; print(json.dumps({k:u.get(k) for k in ('value','decode_ms_per_step','prefill_ms','error','trace')}))" ) >> gpurun_out/und_batch.log 2>&1
done
done
for ub in 1 4 16; do
  echo "== und-batch $ub" >> gpurun_out/und_batch.log
  ( timeout 300 python bench.py --only-understanding --und-batch $ub 2>&1 | grep '^{' | python -c "import json,sys; u=json.loads(sys.stdin.read())['understanding']; print(json.dumps({k:u.get(k) for k in ('value','decode_ms_per_step','prefill_ms','error','trace')}))" ) >> gpurun_out/und_batch.log 2>&1
done
cat gpurun_out/und_batch.log
