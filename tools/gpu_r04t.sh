#!/bin/bash
# Weight gradients on a second stream: training parity tests, then same-box A/B of config 5.
cd /root/repo
python -m pytest tests/test_generator_train_gpu.py tests/test_encoder_train_gpu.py tests/test_cycle_gpu.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  SMIRK_TRAIN_WGRAD_STREAM=$v python bench.py --workload train64 --steps 12 --warmup 3 --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04t_train64_side$v.json
  echo "wgrad side stream=$v $(python tools/bench_summary.py gpurun_out/r04t_train64_side$v.json 0 | head -1)"
done
