#!/bin/bash
# PMC passes (one counter group per rocprofv3 run, kernel-trace only) over a reduced text->image step that launches the four
# denoise GEMM shapes (M = 16392) in the same proportions as the full run; per-kernel averages -> gpurun_out/pmc_*.txt
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o pmc -- python $ROOT/bench.py --layers 2 --num-timesteps 3 --no-vae --no-understanding --no-cpu-baseline --no-taylorseer --warmup 0 --steps 1 > $ROOT/gpurun_out/pmc_run_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB gemm_p > "$ROOT/gpurun_out/pmc_$(echo $grp | tr ' ' '_').txt" 2>&1
  rm -rf /tmp/pmc_$i
done
cd $ROOT
tail -n 40 gpurun_out/pmc_*.txt
du -sh gpurun_out
