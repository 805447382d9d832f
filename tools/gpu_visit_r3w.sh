#!/bin/bash
# round 3, visit w: vectorised qknorm/rope reverse, two-pass rmsnorm reverse, column-sum timestep gradient: parity + sequence of a 2-layer step
mkdir -p gpurun_out
ROOT=$PWD
timeout 900 python -m pytest tests/test_train_backward_gpu.py -q 2>&1 | tail -6 > gpurun_out/r3w_pytest.log
tail -6 gpurun_out/r3w_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ts
PROBE_LAYERS=2 PROBE_ITERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python $ROOT/tools/train_step_probe.py > $ROOT/gpurun_out/r3w_prof_run.log 2>&1
cd $ROOT
grep "^{" gpurun_out/r3w_prof_run.log | cut -c100-330
DB=$(find /tmp/prof_ts -name "*.db" | head -1)
python tools/rocprof_sequence.py $DB "" 130 > gpurun_out/r3w_sequence.log 2>&1
