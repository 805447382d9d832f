#!/bin/bash
# Round-4 visit 17: counter list of this box + two SQ counter groups over the attention-reverse probe
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 -L > $ROOT/gpurun_out/v17_counters.txt 2>&1
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pmcb_$i
  PROBE_LSE=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcb_$i -o pmc -- python $ROOT/tools/attn_bwd_probe.py > $ROOT/gpurun_out/v17_pmc_run_$i.log 2>&1
  DB=$(find /tmp/pmcb_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB > "$ROOT/gpurun_out/v17_pmc_bwd_$i.txt" 2>&1
  rm -rf /tmp/pmcb_$i
done
cd $ROOT
grep -A12 "attn_bwd" gpurun_out/v17_pmc_bwd_1.txt | head -40; grep -A12 "attn_bwd" gpurun_out/v17_pmc_bwd_2.txt | head -40
grep -c . gpurun_out/v17_counters.txt
