#!/bin/bash
TAG=${1:-r04d}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_render_gpu.py tests/test_generator_gpu.py -q -x -k "enc1 or straddling or generator" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -15 $OUT/${TAG}_pytest.log
timeout 300 python tools/enc1_bench.py --batches 128,1024 --generator 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_enc1_bench.txt
