#!/bin/bash
# First GPU visit of the next round: evaluate the opt-in variants that were written without a GPU (round 1 ran out of
# GPU-minutes): BAGEL_ATTN_SCHED=1 (attention.hip), BAGEL_CFG_BATCH=1 (+ BAGEL_UND_SIDE) (bagel.py / qwen2_navit.py).
# Everything is wrapped in timeouts; logs under gpurun_out/.  ~15 GPU-minutes.  (attn_probe runs each schedule in its own child
# process under a 600 s timeout: a variant that hangs cannot take the visit down with it.)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--steps 1 --warmup 1 --no-understanding --no-taylorseer --no-cpu-baseline"
( timeout 300 python tools/attn_probe.py --compare ) > gpurun_out/attn_sched.log 2>&1; tail -8 gpurun_out/attn_sched.log
for sv in 1 2 3 4 5; do ( BAGEL_ATTN_SCHED=$sv timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 300 ) > gpurun_out/pytest_attn_sched$sv.log 2>&1; tail -4 gpurun_out/pytest_attn_sched$sv.log; done
( BAGEL_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_experimental_gpu.py -m gpu -q --timeout 800 ) > gpurun_out/pytest_experimental.log 2>&1; tail -4 gpurun_out/pytest_experimental.log
( timeout 900 python tools/check_stream_batch.py --bench ) > gpurun_out/stream_batch.log 2>&1; tail -32 gpurun_out/stream_batch.log
( BAGEL_CFG_BATCH=1 timeout 600 python -m pytest tests/test_model_gpu.py tests/test_inferencer_gpu.py -m gpu -q --timeout 300 ) > gpurun_out/pytest_cfg_batch.log 2>&1; tail -4 gpurun_out/pytest_cfg_batch.log
# (the relative comparison of all variants is the in-process sweep of check_stream_batch.py --bench above: one 7B model build,
# 5 Euler steps per mode)
# ... then the full bench line for the default and for the combination expected to win (re-run with the measured best if it differs)
for v in "" "BAGEL_ATTN_SCHED=2 BAGEL_CFG_BATCH=1"; do
  tag=$(echo "${v:-default}" | tr ' =' '__')
  ( env $v timeout 600 python bench.py $B ) > gpurun_out/bench_$tag.log 2>&1
  tail -1 gpurun_out/bench_$tag.log | cut -c1-400
done
du -sh gpurun_out
