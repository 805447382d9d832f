#!/bin/bash
# round 3, visit D: the whole GPU suite with the planned attention kernel as the engine's default, then same-box A/B of the bench's denoise
# and edit workloads with the one-tile-per-workgroup kernel (BAGEL_ATTN_PLANNED=0) and the planned one
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 900 ) > gpurun_out/r3d_pytest.log 2>&1
tail -8 gpurun_out/r3d_pytest.log
X="--steps 1 --warmup 1 --no-taylorseer --no-fp8 --no-edit --no-understanding --no-cpu-baseline"
for wl in t2i edit; do for pl in 0 1; do
  echo "== workload=$wl planned=$pl"
  BAGEL_ATTN_PLANNED=$pl timeout 600 python bench.py $X --workload $wl 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['achieved'], d.get('whole_path_roofline',{}).get('frac'))"
done; done
