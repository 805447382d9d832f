#!/bin/bash
# ad-hoc GPU visit: fused decode attention parity + speed, bench with the full-size parity probe.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_model_gpu.py -q --timeout 300 ) > gpurun_out/pytest.log 2>&1
tail -12 gpurun_out/pytest.log
for v in "BAGEL_DECODE_FUSED=1" "BAGEL_DECODE_FUSED=0"; do
  echo "== $v" >> gpurun_out/fused.log
  ( env $v timeout 300 python bench.py --only-understanding 2>&1 | grep '^{' | python -c "import json,sys; u=json.loads(sys.stdin.read())['understanding']; print(u.get('value'), u.get('decode_ms_per_step'), u.get('error'))" ) >> gpurun_out/fused.log 2>&1
done
cat gpurun_out/fused.log
( time timeout 900 python bench.py --steps 1 --warmup 1 --no-taylorseer ) > gpurun_out/bench.log 2>&1
tail -2 gpurun_out/bench.log
