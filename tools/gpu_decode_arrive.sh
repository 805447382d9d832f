#!/bin/bash
# One-launch decode attention (last-arriver merge) vs split + combine: parity tests, kernel-level timing, whole understanding leg.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_und_shapes_gpu.py -m gpu -q -x --timeout 600 ) > gpurun_out/arrive_pytest.log 2>&1; tail -3 gpurun_out/arrive_pytest.log | cut -c1-300
for rep in 1 2; do
  for ar in 0 1; do
    for B in 1 2 16; do
      BAGEL_DECODE_ARRIVE=$ar timeout 300 python tools/attn_decode_bench.py $B 2>&1 | grep -v amdgpu | tail -1
    done
  done
done > gpurun_out/arrive_attn_bench.log 2>&1
cat gpurun_out/arrive_attn_bench.log
for rep in 1 2; do
  for ar in 0 1; do
    ( BAGEL_DECODE_ARRIVE=$ar timeout 900 python bench.py --only-understanding --no-cpu-baseline --no-int8 ) 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())['understanding']
bd=d.get('batched_decode') or {}
print('ARRIVE=$ar', 'B=1 tok/s %.1f  ms/step %.3f  prefill %s' % (d['value'], d['decode_ms_per_step'], d['prefill_ms']), '| batched:', json.dumps(bd)[:400])
"
  done
done > gpurun_out/arrive_und.log 2>&1
cat gpurun_out/arrive_und.log
