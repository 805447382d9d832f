export TMPDIR=/tmp
for a in 7 5 0; do echo "#### ABL=$a"; BAGEL_ENGINE_ABL=$a python tools/decode_engine_probe.py --trace --sets 2 2>&1 | grep -v "amdgpu.ids\|    c1 \|    c2 " | sed 's/  p10.*med/  med/; s/  p90.*max/  max/' | cut -c1-100; done
