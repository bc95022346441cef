#!/bin/bash
# round-3 call N: full-line epilogue stores in the halo-patch kernels — parity + A/B (variant library built with -DSMIRK_PATCH_COALESCE=0)
TAG=${1:-r03n}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_generator_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${TAG}_pytest.log | cut -c1-200
for B in 256 1024; do
  echo "== B=$B coalesced (default)" >> $OUT/${TAG}_patch_ab.txt
  timeout 300 python tools/conv_sweep.py --batch $B --iters 5 2>&1 | grep -E "enc1|dec1|enc2|dec2" >> $OUT/${TAG}_patch_ab.txt
  echo "== B=$B per-lane hi/lo stores (variant)" >> $OUT/${TAG}_patch_ab.txt
  SMIRK_HIP_LIBRARY=/root/repo/smirk_amd/lib_fz/libsmirk_hip_variant.so timeout 300 python tools/conv_sweep.py --batch $B --iters 5 2>&1 | grep -E "enc1|dec1|enc2|dec2" >> $OUT/${TAG}_patch_ab.txt
done
cut -c1-120 $OUT/${TAG}_patch_ab.txt
