#!/bin/bash
# round 3, visit z: tape keeps gate/up + cached transposed weights: parity, full-depth step in both modes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_backward_gpu.py -q 2>&1 | tail -4 > gpurun_out/r3z_pytest.log
tail -4 gpurun_out/r3z_pytest.log
for k in 0 1; do
  BAGEL_TRAIN_KEEP_GATE_UP=$k PROBE_ITERS=2 timeout 900 python tools/train_step_probe.py 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('keep_gate_up=$k', {x: round(d[x],1) for x in ('ms_forward','ms_backward','tokens_per_s','max_mem_gb')})"
done
PROBE_ITERS=2 timeout 900 python tools/train_step_probe.py 2>&1 | grep "^{" > gpurun_out/r3z_probe28.log; cut -c1-400 gpurun_out/r3z_probe28.log
