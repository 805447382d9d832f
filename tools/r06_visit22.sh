#!/bin/bash
# round 6, visit 22: PMC passes (FETCH_SIZE, WRITE_SIZE; one group per run, --kernel-trace only) over a 32-request decode: HBM-side bytes per step of the two-block weight stream
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcb_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcb_$i -o pmc -- python $ROOT/bench.py --only-understanding --und-batch 32 --und-new-tokens 24 --no-cpu-baseline --no-int8 --no-batched-decode > $ROOT/gpurun_out/pmc_decode32_run_$i.log 2>&1
  DB=$(find /tmp/pmcb_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB > "$ROOT/gpurun_out/pmc_decode32_$grp.txt" 2>&1
  rm -rf /tmp/pmcb_$i
done
cd $ROOT
grep -A1 "gemv_mb\|attn_decode\|argmax" gpurun_out/pmc_decode32_FETCH_SIZE.txt | head -40
