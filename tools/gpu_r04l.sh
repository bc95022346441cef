#!/bin/bash
cd /root/repo
python -m pytest tests/test_generator_gpu.py tests/test_scale_gpu.py -q -x 2>&1 | tail -2
for gb in 128 1024 256; do
  for sc in 1 0; do
    SMIRK_GEN_SPLIT_CHAINS=$sc python bench.py --workload full --global-batch $gb --force-collective --steps 30 --warmup 5 --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04l_b${gb}_sc${sc}.json
    echo "gb=$gb split_chains=$sc $(python tools/bench_summary.py gpurun_out/r04l_b${gb}_sc${sc}.json 0 | head -1)"
  done
done
