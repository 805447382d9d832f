#!/bin/bash
# Round-4 visit 3: phase timing of a batched generate_text call; the 7B-width image-edit parity test; the training tests after the
# engine-refresh / tape-pool rework; the training-step probe with an in-place optimizer bump.
mkdir -p gpurun_out
export TMPDIR=/tmp
( BAGEL_DEC_CPW=2 timeout 900 python tools/decode_phase_probe.py 16 160 ) > gpurun_out/v3_phase_b16.log 2>&1; grep "^rep" gpurun_out/v3_phase_b16.log
( timeout 900 python tools/decode_phase_probe.py 2 160 ) > gpurun_out/v3_phase_b2.log 2>&1; grep "^rep" gpurun_out/v3_phase_b2.log
( timeout 900 python -m pytest tests/test_wide_gpu.py -m gpu -q -x --timeout 600 -k "image_edit or text_to_image" -s ) > gpurun_out/v3_pytest_wide.log 2>&1; grep -E "parity|passed|failed|Error" gpurun_out/v3_pytest_wide.log | cut -c1-600
( timeout 1200 python -m pytest tests/test_train_backward_gpu.py tests/test_train_gpu.py -m gpu -q -x --timeout 900 ) > gpurun_out/v3_pytest_train.log 2>&1; tail -3 gpurun_out/v3_pytest_train.log
( timeout 900 python -m pytest tests/test_model_gpu.py tests/test_inferencer_gpu.py tests/test_variants_gpu.py -m gpu -q --timeout 600 ) > gpurun_out/v3_pytest_model.log 2>&1; tail -3 gpurun_out/v3_pytest_model.log
find gpurun_out -size +5M -delete
