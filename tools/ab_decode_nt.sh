for lib in "" plain ntw "" plain ntw; do
  if [ -n "$lib" ]; then export BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$lib.so; else unset BAGEL_HIP_LIB; fi
  echo "lib=${lib:-default(all nt)}"
  timeout 300 python bench.py --only-understanding --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['understanding']; print('  B1', round(d['value'],1), round(d['decode_ms_per_token'],3), 'int8', round(d['int8_rowwise_weights']['value'],1))"
done
for lib in "" plain; do
  if [ -n "$lib" ]; then export BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$lib.so; else unset BAGEL_HIP_LIB; fi
  timeout 300 python bench.py --only-understanding --no-cpu-baseline --und-batch 8 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['understanding']; print('  B8 ${lib:-default}', round(d['value'],1), round(d['decode_ms_per_step'],3))"
done
