#!/bin/bash
# round 6, third GPU pass: BN kernels restored, pack-plan kinds, fused statistics A/B, phase times of the training step
TAG=${1:-r06c}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_train_ops_gpu.py tests/test_generator_train_gpu.py tests/test_encoder_train_gpu.py tests/test_cycle_gpu.py tests/test_train_scale_gpu.py -m gpu -q -x > $OUT/${TAG}_pytest_part.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_part.log; tail -12 $OUT/${TAG}_pytest_part.log | cut -c1-250
timeout 200 python tools/train_phase_times.py f16x3 8 2>&1 | tail -10 | tee $OUT/${TAG}_phase_times_f16x3.txt
timeout 200 python tools/train_phase_times.py f16x1 8 2>&1 | tail -10 | tee $OUT/${TAG}_phase_times_f16x1.txt
for rep in 1 2; do
  timeout 300 python bench.py --workload train64 --steps 20 --warmup 3 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_fused$rep.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_fused$rep.json 1
  SMIRK_BN_STATS_UNFUSED=1 timeout 300 python bench.py --workload train64 --steps 20 --warmup 3 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_unfused$rep.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_unfused$rep.json 1
done
timeout 300 python bench.py --workload train64 --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_bench_train64.err; python tools/bench_summary.py $OUT/${TAG}_bench_train64.json 30
