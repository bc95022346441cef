#!/bin/bash
# The chained section starting at H/4 (level 2) vs H/8 (level 3): parity with the chains forced, then same-box A/B at 128 / 256 / 1024 frames per pass.
cd /root/repo
for from in 2 3; do
  for sc in 2 3; do
    echo "== tests with SMIRK_GEN_SPLIT_CHAINS=$sc SMIRK_GEN_CHAIN_FROM=$from"
    SMIRK_GEN_SPLIT_CHAINS=$sc SMIRK_GEN_CHAIN_FROM=$from python -m pytest tests/test_generator_gpu.py tests/test_scale_gpu.py -q -x 2>&1 | tail -2
  done
done
for gb in 128 256 1024; do
  for cfg in "2 2" "2 3" "0 3" "3 2"; do
    set -- $cfg
    SMIRK_GEN_SPLIT_CHAINS=$1 SMIRK_GEN_CHAIN_FROM=$2 python bench.py --workload full --global-batch $gb --force-collective --steps 30 --warmup 5 --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04n_b${gb}_sc$1_from$2.json
    echo "gb=$gb chains=$1 from=$2 $(python tools/bench_summary.py gpurun_out/r04n_b${gb}_sc$1_from$2.json 0 | head -1)"
  done
done
