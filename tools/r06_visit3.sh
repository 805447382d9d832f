#!/bin/bash
# round 6, visit 3: attention DMA-path A/B + timing ablations, parity of the fast DMA path, edit batch sizes, SigLIP fc2 padding
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2; do for lib in "" slowdma abl8 abl2 abl10; do
  if [ -n "$lib" ]; then export BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$lib.so; else unset BAGEL_HIP_LIB; fi
  echo "== lib=${lib:-default(fast dma)} round $r"
  timeout 200 python tools/attn2_probe.py --iters 30 2>&1 | grep -E "denoise_b8|edit_3|prefill_4936|siglip" | cut -c1-175
done; done > gpurun_out/v3_attn_ab.log 2>&1
unset BAGEL_HIP_LIB
timeout 900 python -m pytest tests/test_attn2_gpu.py -x -q > gpurun_out/v3_attn2_tests.log 2>&1
echo "attn2 tests rc=$?" > gpurun_out/v3_rc.txt
for nb in 3 4; do
  timeout 600 python bench.py --no-understanding --no-cpu-baseline --no-taylorseer --no-fp8 --no-train-forward --warmup 0 --edit-batch $nb > gpurun_out/v3_edit_b$nb.log 2> gpurun_out/v3_edit_b$nb.err
  echo "edit batch $nb rc=$?" >> gpurun_out/v3_rc.txt
done
timeout 600 python bench.py --only-understanding --no-cpu-baseline --no-int8 --no-batched-decode > gpurun_out/v3_und.log 2> gpurun_out/v3_und.err
echo "und rc=$?" >> gpurun_out/v3_rc.txt
cat gpurun_out/v3_rc.txt; cat gpurun_out/v3_attn_ab.log
