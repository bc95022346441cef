#!/bin/bash
# First GPU run of the f16x1 training arithmetic (branch wip/train-f16x1): the op-level and whole-network parity tests, then config 5 in both arithmetics.
cd /root/repo
timeout 150 python -m pytest tests/test_train_ops_gpu.py tests/test_generator_train_gpu.py -q -x -k "f16x1" -s 2>&1 | grep -v "^$" | tail -8
for ar in f16x1 f16x3; do
  timeout 120 python bench.py --workload train64 --train-arith $ar --steps 8 --warmup 3 --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04y_train64_$ar.json
  echo "train_arith=$ar $(python tools/bench_summary.py gpurun_out/r04y_train64_$ar.json 0 | head -1)"
done
