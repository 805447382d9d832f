#!/bin/bash
# round 6, visit 12b: gemv_mb at 16 / 32 rows with fewer workgroups (every workgroup re-reads ALL activation rows: 229 KB at 32 rows against 114 KB of weights per column block)
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in 256 128 64; do
  echo "== BAGEL_MB_WGS=$w" >> gpurun_out/v12b_mb_wgs.log
  ( BAGEL_MB_WGS=$w timeout 600 python tools/gemv_mb_bench.py 16 32 ) >> gpurun_out/v12b_mb_wgs.log 2>&1
done
cat gpurun_out/v12b_mb_wgs.log
