#!/bin/bash
# PMC passes over tools/decode_engine_probe.py --quick (launch form: gemv_kernel<1,1> / <1,4>; engine: decode_engine_kernel), one counter group per
# rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md recipe).  -> gpurun_out/pmc_engine_<n>.txt
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
cd /tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmce_$i
  ( timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmce_$i -o pmc -- python $ROOT/tools/decode_engine_probe.py --quick --reps 3 --sets 3 ) > $ROOT/gpurun_out/pmc_engine_run_$i.log 2>&1
  DB=$(find /tmp/pmce_$i -name "*.db" | head -1)
  echo "## counters: $grp" > $ROOT/gpurun_out/pmc_engine_$i.txt
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB >> $ROOT/gpurun_out/pmc_engine_$i.txt 2>&1
  grep -A12 "gemv_kernel\|decode_engine_kernel" $ROOT/gpurun_out/pmc_engine_$i.txt | grep -v "^--" | head -60
  tail -3 $ROOT/gpurun_out/pmc_engine_run_$i.log | cut -c1-200
  rm -rf /tmp/pmce_$i
done
