#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_nf4_gpu.py -m gpu -q --timeout 600 -x ) 2>&1 | tail -25
( timeout 900 python bench.py --only-understanding --no-cpu-baseline ) 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())['understanding']
for k in ('value','prefill_ms','decode_ms_per_token','int8_weights','mxfp4_weights','nf4_weights','batched_decode'): print(k, d.get(k))
print(d.get('roofline'))"
