#!/bin/bash
TAG=${1:-r03j}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
run() { name=$1; shift; timeout 600 python -m pytest "$@" -q -x -p no:cacheprovider > $OUT/${TAG}_$name.log 2>&1; echo "$name rc=$?  $(grep -E 'passed|failed' $OUT/${TAG}_$name.log | tail -1 | cut -c1-120)"; }
run all_k tests -m gpu -k "conv_gpu or cycle_gpu or test_conv or test_cycle or graphed"
run full tests -m gpu
