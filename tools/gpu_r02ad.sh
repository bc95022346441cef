#!/bin/bash
TAG=${1:-r02ad}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_train_ops_gpu.py -q -x -k "weight_gradient" > $OUT/${TAG}_pytest_wgrad.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_wgrad.log; tail -3 $OUT/${TAG}_pytest_wgrad.log
for m in 1 2; do SMIRK_WGRAD_F16=$m python tools/wgrad_sweep.py 64 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_wgrad_sweep_mode$m.txt | tail -23; done
