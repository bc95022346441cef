#!/bin/bash
TAG=${1:-r04h}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_generator_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -12 $OUT/${TAG}_pytest.log
for B in 1024 128; do
  echo "== B=$B ring (default)"; timeout 200 python tools/conv_sweep.py --batch $B --iters 5 --only "1" 2>&1 | grep "enc1\|dec1"; timeout 200 python tools/conv_sweep.py --batch $B --iters 5 --only "2" 2>&1 | grep "enc2\|dec2"
  echo "== B=$B SMIRK_CONV_RING=0"; SMIRK_CONV_RING=0 timeout 200 python tools/conv_sweep.py --batch $B --iters 5 --only "dec1a" 2>&1 | grep "dec1a"
done | tee $OUT/${TAG}_ring_sweep.txt
timeout 300 python tools/enc1_bench.py --batches 128,1024 --generator 2>&1 | grep generator
