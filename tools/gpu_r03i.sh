#!/bin/bash
# round-3 call I: bisect the capture_end segfault that only shows with whole-directory collection
TAG=${1:-r03i}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
run() { name=$1; shift; timeout 300 python -m pytest "$@" -q -x -p no:cacheprovider > $OUT/${TAG}_$name.log 2>&1; echo "$name rc=$?  $(grep -E 'passed|failed' $OUT/${TAG}_$name.log | tail -1 | cut -c1-120)"; }
run all_k tests -m gpu -k "conv_gpu or cycle_gpu or test_conv or test_cycle or graphed"
run ign_rccl tests -m gpu -k "test_conv or test_cycle or graphed" --ignore tests/test_rccl_gpu.py
run ign_isa tests -m gpu -k "test_conv or test_cycle or graphed" --ignore tests/test_isa_cpu.py
run ign_trainscale tests -m gpu -k "test_conv or test_cycle or graphed" --ignore tests/test_train_scale_gpu.py
run ign_three tests -m gpu -k "test_conv or test_cycle or graphed" --ignore tests/test_train_scale_gpu.py --ignore tests/test_isa_cpu.py --ignore tests/test_rccl_gpu.py
run cycle_only_dir tests -m gpu -k "test_cycle or graphed"
run capture_only_dir tests -m gpu -k "graphed"
