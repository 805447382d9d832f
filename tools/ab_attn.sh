# same-box A/B of attention kernel build variants (tools/ab_build.sh): denoise-shape probe, two rounds
for r in 1 2; do for lib in "" $*; do
  if [ -n "$lib" ]; then export BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_$lib.so; else unset BAGEL_HIP_LIB; fi
  echo -n "lib=${lib:-default}  "; timeout 120 python tools/attn_probe.py 2>&1 | grep attn_denoise
done; done
