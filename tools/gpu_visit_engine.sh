#!/bin/bash
# Record visit for the persistent decode engine: parity tests, the projection-chain probe (engine vs launch form, knob sweep + timing ablations), one traced
# launch, then the understanding leg of bench.py with the engine off and on.  Logs -> gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_engine.log 2>&1
tail -3 gpurun_out/pytest_engine.log
( timeout 600 python tools/decode_engine_probe.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/engine_probe.log
cat gpurun_out/engine_probe.log
( timeout 600 python tools/decode_engine_probe.py --trace --sets 2 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/engine_trace.log
for e in 0 1; do
  ( BAGEL_DECODE_ENGINE=$e timeout 600 python bench.py --only-understanding --no-int8 --no-cpu-baseline --no-batched-decode --und-new-tokens 128 ) > gpurun_out/und_engine$e.log 2>&1
  echo "== BAGEL_DECODE_ENGINE=$e"; grep -o '"decode_ms_per_token": [0-9.]*\|"hip_graph_error": [^,]*' gpurun_out/und_engine$e.log | head -4
done
