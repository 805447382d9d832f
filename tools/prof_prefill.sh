export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_pf -o und -- python $ROOT/bench.py --only-understanding --no-cpu-baseline --no-int8 --und-new-tokens 8 > $ROOT/gpurun_out/pf_prof.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_pf -name "*.db" | head -1)
python tools/rocprof_by_grid.py $DB attn_fwd
python tools/rocprof_by_grid.py $DB gemm_p | head -24
python tools/rocprof_summary.py $DB 2>/dev/null | head -24
rm -rf gpurun_out/prof_pf
