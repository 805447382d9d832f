# per-shape timing of the MXFP4 decode projections under rocprofv3: variants = environment settings / A-B libraries given as arguments
export TMPDIR=/tmp
ROOT=$PWD
i=0
for v in "A=1" "$@"; do
  i=$((i+1))
  cd /tmp
  env $v timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_w4_$i -o und -- python $ROOT/bench.py --only-understanding --no-cpu-baseline --und-new-tokens 64 > $ROOT/gpurun_out/w4_prof_$i.log 2>&1
  cd $ROOT
  DB=$(find gpurun_out/prof_w4_$i -name "*.db" | head -1)
  echo "== $v"; grep -o '"mxfp4_weights": {"value": [0-9.]*' gpurun_out/w4_prof_$i.log
  python tools/rocprof_by_grid.py $DB gemv_w4 | tail -4
  rm -rf gpurun_out/prof_w4_$i
done
