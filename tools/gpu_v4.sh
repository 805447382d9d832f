#!/bin/bash
# Round-4 visit 4: gemv_mb after the geometry chooser (tests + per-shape timing), decode attention alone over chunks-per-workgroup,
# kernel stats of the 16-request decode step.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
( timeout 600 python -m pytest tests/test_gemv_mb_gpu.py -m gpu -q -x --timeout 300 ) > gpurun_out/v4_pytest_mb.log 2>&1; tail -2 gpurun_out/v4_pytest_mb.log
( timeout 600 python tools/gemv_mb_bench.py 2 16 ) > gpurun_out/v4_gemv_mb_bench.log 2>&1; grep -v amdgpu gpurun_out/v4_gemv_mb_bench.log
: > gpurun_out/v4_attn_decode.log
for cpw in 1 2 3 4 5 6 8; do ( BAGEL_DEC_CPW=$cpw timeout 200 python tools/attn_decode_bench.py 16 ) >> gpurun_out/v4_attn_decode.log 2>&1; done
for cpw in 1 2; do ( BAGEL_DEC_CPW=$cpw timeout 200 python tools/attn_decode_bench.py 2 ) >> gpurun_out/v4_attn_decode.log 2>&1; ( BAGEL_DEC_CPW=$cpw timeout 200 python tools/attn_decode_bench.py 1 ) >> gpurun_out/v4_attn_decode.log 2>&1; done
grep -v amdgpu gpurun_out/v4_attn_decode.log
cd /tmp
( BAGEL_DEC_CPW=2 timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_b16 -o und -- python $ROOT/tools/decode_phase_probe.py 16 64 ) > $ROOT/gpurun_out/v4_b16_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_b16 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > gpurun_out/v4_b16_kernel_stats.csv 2>gpurun_out/v4_err.log
grep -E "gemv_mb|attn_decode|reduce|argmax|copy_rows|rope_table|advance" gpurun_out/v4_b16_kernel_stats.csv | cut -c1-110
rm -rf gpurun_out/prof_b16
find gpurun_out -size +5M -delete
