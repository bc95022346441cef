#!/bin/bash
TAG=${1:-r04e}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_render_gpu.py tests/test_generator_gpu.py -q -x -k "enc1 or straddling or generator" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -12 $OUT/${TAG}_pytest.log
for B in 128 1024; do SMIRK_HIP_LIBRARY=/root/repo/smirk_amd/lib_fz/libsmirk_hip_variant.so timeout 120 python tools/enc1_timeline.py $B 2>&1 | grep -v amdgpu; done | tee $OUT/${TAG}_enc1_timeline.txt
