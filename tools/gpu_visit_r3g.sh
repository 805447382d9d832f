#!/bin/bash
# kernel-level A/B inside the bench: rocprofv3 kernel stats of one denoise step with the tile attention kernel and with the planned one
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
X="--steps 1 --warmup 0 --no-taylorseer --no-fp8 --no-edit --no-understanding --no-cpu-baseline --num-timesteps 6"
for pl in 0 1; do
  cd /tmp
  ( BAGEL_ATTN_PLANNED=$pl timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof$pl -o bench -- python $ROOT/bench.py $X ) > $ROOT/gpurun_out/r3g_prof$pl.log 2>&1
  cd $ROOT
  DB=$(find gpurun_out/prof$pl -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/r3g_stats$pl.csv 2>>gpurun_out/r3g_prof$pl.log
  echo "== planned=$pl"; grep -i "attn\|kernel,calls\|v_transpose" gpurun_out/r3g_stats$pl.csv | cut -c1-160
  rm -rf gpurun_out/prof$pl
done
