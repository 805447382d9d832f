set -x
timeout 300 python tools/mall_probe.py > gpurun_out/mall_probe.log 2>&1
BAGEL_HIP_LIB=$PWD/bagel_amd/libbagel_hip_plain.so timeout 300 python tools/mall_probe.py >> gpurun_out/mall_probe.log 2>&1
cat gpurun_out/mall_probe.log
