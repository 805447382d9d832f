#!/bin/bash
# PMC passes (rounds 2-6) (one counter group per rocprofv3 run, --kernel-trace only: gpurun refuses --pmc with the other trace domains),
# exactly as MI355X_MICROARCH.md prescribes:
#   (1) denoise: a reduced text->image step (2 layers, 3 timesteps, default execution = stream-batched CFG + marker side path) that
#       launches the four gen-expert GEMM shapes at M = 32 768 and the attention kernel on 8 x 4098 rows in the proportions of the
#       full run                                              -> gpurun_out/pmc_denoise_<group>.txt
#   (2) decode: bench.py --only-understanding, 24 new tokens  -> gpurun_out/pmc_decode_<group>.txt
# then tools/pmc_make_summary.py folds the per-kernel averages into profiles/r06_pmc_summary.json (stamped with the git commit and the
# sha1 of the kernel sources, which bench.py checks before quoting `traffic`).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
[ "$ONLY" = "decode" ] || for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o pmc -- python $ROOT/bench.py --layers 2 --num-timesteps 3 --no-vae --no-understanding --no-cpu-baseline --no-taylorseer --no-edit --no-fp8 --no-train-forward --warmup 0 --steps 1 > $ROOT/gpurun_out/pmc_denoise_run_$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB > "$ROOT/gpurun_out/pmc_denoise_$tag.txt" 2>&1
  rm -rf /tmp/pmc_$i
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcd_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcd_$i -o pmc -- python $ROOT/bench.py --only-understanding --und-new-tokens 24 --no-cpu-baseline --no-int8 --no-batched-decode > $ROOT/gpurun_out/pmc_decode_run_$i.log 2>&1
  DB=$(find /tmp/pmcd_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $ROOT/tools/pmc_summary.py $DB > "$ROOT/gpurun_out/pmc_decode_$grp.txt" 2>&1
  rm -rf /tmp/pmcd_$i
done
cd $ROOT
python tools/pmc_make_summary.py gpurun_out gpurun_out/r06_pmc_summary.json
head -c 3000 gpurun_out/r06_pmc_summary.json
du -sh gpurun_out
