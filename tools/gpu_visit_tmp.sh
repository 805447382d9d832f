export TMPDIR=/tmp
python -m pytest tests/test_wide_gpu.py -m gpu -q --timeout 600 -k "leftover or gemm_persistent" 2>&1 | tail -2
for pol in 0 1; do
  echo "== BAGEL_GEMM_SPLIT_POLICY=$pol"
  BAGEL_GEMM_SPLIT_POLICY=$pol python bench.py --workload edit --steps 1 --warmup 1 --no-cpu-baseline --no-understanding --no-taylorseer --no-fp8 --no-train-forward 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('edit', d.get('value'), d.get('unit'), d.get('ms_per_step'))"
  BAGEL_GEMM_SPLIT_POLICY=$pol python bench.py --only-understanding --no-int8 --no-batched-decode --no-cpu-baseline --und-new-tokens 16 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('prefill', d['understanding']['prefill_ms'])"
done
