#!/bin/bash
# round 6, visit 4: FP8 delayed scales (kernel test, model tests, bench A/B), SigLIP fc2 shape
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fp8_gpu.py -x -q -s > gpurun_out/v4_fp8_tests.log 2>&1
echo "fp8 tests rc=$?" > gpurun_out/v4_rc.txt
timeout 900 python -m pytest tests/test_wide_gpu.py -x -q -s -k fp8 > gpurun_out/v4_wide_fp8.log 2>&1
echo "wide fp8 rc=$?" >> gpurun_out/v4_rc.txt
timeout 300 python tools/gemm_prefill_shapes.py 2>&1 | grep -i "vit" > gpurun_out/v4_vit_shapes.log
for dl in 1 0; do
  BAGEL_FP8_DELAYED=$dl timeout 600 python bench.py --no-understanding --no-cpu-baseline --no-taylorseer --no-edit --no-train-forward --warmup 1 > gpurun_out/v4_fp8_delayed$dl.log 2> gpurun_out/v4_fp8_delayed$dl.err
  echo "bench delayed=$dl rc=$?" >> gpurun_out/v4_rc.txt
done
cat gpurun_out/v4_rc.txt; tail -5 gpurun_out/v4_fp8_tests.log; cat gpurun_out/v4_vit_shapes.log
