#!/bin/bash
# Round-4 visit 12: where does the first Euler step of the 49-step trajectory pick up 1.75 x the reference's own noise?
mkdir -p gpurun_out
export TMPDIR=/tmp
for envs in "" "BAGEL_CFG_BATCH=0" "BAGEL_UND_SIDE=0" "BAGEL_ATTN_PLANNED=0" "BAGEL_GEMM_VARIANT=0" "BAGEL_GEMV_MB=0" "BAGEL_CFG_BATCH=0 BAGEL_UND_SIDE=0 BAGEL_ATTN_PLANNED=0 BAGEL_GEMM_VARIANT=0 BAGEL_GEMV_MB=0"; do
  ( env $envs timeout 600 python tools/traj_probe.py ) 2>&1 | grep -E "^rep|Error|error" | cut -c1-600
done > gpurun_out/v12_traj_probe.log 2>&1
cat gpurun_out/v12_traj_probe.log
