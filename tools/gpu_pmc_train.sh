#!/bin/bash
# PMC passes over a 2-layer training step (tools/train_step_probe.py, SigLIP tower frozen): one counter group per rocprofv3 run,
# --kernel-trace only (gpurun refuses --pmc with the other trace domains), as MI355X_MICROARCH.md prescribes.
#   gpurun_out/pmc_train_<group>.txt = per-kernel averages (tools/pmc_summary.py)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  [ -n "$ONLY_GROUPS" ] && ! echo " $ONLY_GROUPS " | grep -q " $i " && continue
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmct_$i
  PROBE_LAYERS=2 PROBE_ITERS=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmct_$i -o pmc -- python $ROOT/tools/train_step_probe.py > $ROOT/gpurun_out/pmc_train_run_$i.log 2>&1
  DB=$(find /tmp/pmct_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB > "$ROOT/gpurun_out/pmc_train_$tag.txt" 2>&1
  rm -rf /tmp/pmct_$i
done
cd $ROOT
ls -la gpurun_out/pmc_train_*.txt
