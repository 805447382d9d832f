#!/bin/bash
# round 3, visit A: the new parity tests at the understanding / VAE shapes, the Level-2 seam, the depth-4 Euler step; then the bench's
# CPU leg with the full-depth (28-layer) Euler step through the oracle.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_und_shapes_gpu.py tests/test_seam_gpu.py tests/test_full_depth_gpu.py -m gpu -q -x --timeout 600 -s ) > gpurun_out/r3a_pytest.log 2>&1
tail -15 gpurun_out/r3a_pytest.log
nproc; free -g | head -2
( time timeout 1500 python bench.py --steps 1 --warmup 1 --no-taylorseer --no-fp8 --no-edit --no-understanding ) > gpurun_out/r3a_bench.log 2>&1
tail -1 gpurun_out/r3a_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ranks_seen','rccl_version','broadcast')}); print(json.dumps(d['cpu_baseline'],indent=1)[:3000])"
