#!/bin/bash
# SQ / LDS counters of the attention-reverse probe on whatever library BAGEL_HIP_LIB names (default: product)
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
LIBV=${LIBV:-dkvnew}
cd /tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmcb_$i
  BAGEL_HIP_LIB=$ROOT/bagel_amd/libbagel_hip_$LIBV.so BAGEL_ABWD_ONLY=dkv PROBE_LSE=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcb_$i -o pmc -- python $ROOT/tools/attn_bwd_probe.py > $ROOT/gpurun_out/bwdpmc_run_$i.log 2>&1
  DB=$(find /tmp/pmcb_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB attn_bwd_dkv > "$ROOT/gpurun_out/bwdpmc_${LIBV}_$i.txt" 2>&1
  rm -rf /tmp/pmcb_$i
  cat $ROOT/gpurun_out/bwdpmc_${LIBV}_$i.txt
done
