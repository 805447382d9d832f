#!/bin/bash
# PMC passes over tools/attn_probe.py (the denoise-shape attention launch) for one kernel variant: where do the wave cycles go?
#   bash tools/attn_pmc.sh <BAGEL_ATTN_KERNEL value> <tag>      -> gpurun_out/attn_pmc_<tag>_<n>.txt
set -x
K=$1; TAG=$2
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/apmc_$i
  BAGEL_ATTN_KERNEL=$K timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/apmc_$i -o pmc -- python $ROOT/tools/attn_probe.py > $ROOT/gpurun_out/attn_pmc_${TAG}_run_$i.log 2>&1
  DB=$(find /tmp/apmc_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB attn_fwd > "$ROOT/gpurun_out/attn_pmc_${TAG}_$i.txt" 2>&1
  rm -rf /tmp/apmc_$i
done
cd $ROOT
cat gpurun_out/attn_pmc_${TAG}_*.txt
