#!/bin/bash
cd /root/repo
SMIRK_GEN_SPLIT_CHAINS=4 python -m pytest tests/test_generator_gpu.py tests/test_scale_gpu.py -q -x 2>&1 | tail -1
SMIRK_GEN_SPLIT_CHAINS=3 python -m pytest tests/test_scale_gpu.py -q -x 2>&1 | tail -1
for gb in 128 256 1024; do
  for sc in 0 2 3 4; do
    SMIRK_GEN_SPLIT_CHAINS=$sc python bench.py --workload full --global-batch $gb --force-collective --steps 30 --warmup 5 --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04m_b${gb}_sc${sc}.json
    echo "gb=$gb chains=$sc $(python tools/bench_summary.py gpurun_out/r04m_b${gb}_sc${sc}.json 0 | head -1)"
  done
done
