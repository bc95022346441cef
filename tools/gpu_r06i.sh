#!/bin/bash
# round 6, batched split-K reductions of the generator's weight gradients (A/B + tests)
TAG=${1:-r06i}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_train_ops_gpu.py tests/test_generator_train_gpu.py tests/test_cycle_gpu.py tests/test_train_scale_gpu.py -m gpu -q -x > $OUT/${TAG}_pytest_part.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_part.log; tail -12 $OUT/${TAG}_pytest_part.log | cut -c1-250
for rep in 1 2; do
  timeout 300 python bench.py --workload train64 --steps 20 --warmup 3 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_$rep.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_$rep.json 1
  SMIRK_WGRAD_UNBATCHED=1 timeout 300 python bench.py --workload train64 --steps 20 --warmup 3 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_unbatched_$rep.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_unbatched_$rep.json 1
done
timeout 300 python bench.py --workload train64 --train-arith f16x1 --steps 20 --warmup 3 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_f16x1.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_f16x1.json 1
timeout 300 python bench.py --workload train64 --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_bench_train64.err; python tools/bench_summary.py $OUT/${TAG}_bench_train64.json 14
