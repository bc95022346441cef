#!/bin/bash
# round-3 call F: locate the encoder mismatch (unit tests of the two new kernels) and the order-dependent capture segfault
TAG=${1:-r03f}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_encoder_gpu.py -q > $OUT/${TAG}_pytest_enc.log 2>&1; echo "encoder pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/${TAG}_pytest_enc.log | cut -c1-250
for v in "SMIRK_DISABLE_MBCONV_IMAGE=1" "SMIRK_DISABLE_ENCODER_HEAD_FUSED=1"; do
  env $v timeout 300 python -m pytest tests/test_encoder_gpu.py -q -k "golden or oracle" > $OUT/${TAG}_pytest_enc_$v.log 2>&1; echo "$v rc=$?"; grep -E "passed|failed" $OUT/${TAG}_pytest_enc_$v.log | cut -c1-200
done
timeout 600 python -m pytest tests/test_cycle_gpu.py -q > $OUT/${TAG}_pytest_cycle_alone.log 2>&1; echo "cycle alone rc=$?"; tail -2 $OUT/${TAG}_pytest_cycle_alone.log | cut -c1-200
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_cycle_gpu.py -q -k "not halo" > $OUT/${TAG}_pytest_conv_cycle_nohalo.log 2>&1; echo "conv(no halo)+cycle rc=$?"; tail -2 $OUT/${TAG}_pytest_conv_cycle_nohalo.log | cut -c1-200
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_cycle_gpu.py -q > $OUT/${TAG}_pytest_conv_cycle.log 2>&1; echo "conv+cycle rc=$?"; tail -2 $OUT/${TAG}_pytest_conv_cycle.log | cut -c1-200
