#!/bin/bash
# Engine iteration visit: parity tests, knob sweep of the projection-chain probe, one traced launch.
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/pytest_engine.log 2>&1
tail -3 gpurun_out/pytest_engine.log
( timeout 600 python tools/decode_engine_probe.py --sets 6 ) > gpurun_out/engine_probe.log 2>&1
cat gpurun_out/engine_probe.log
( timeout 600 python tools/decode_engine_probe.py --trace --sets 2 $TRACE_ENV ) > gpurun_out/engine_trace.log 2>&1
cat gpurun_out/engine_trace.log | tail -64
