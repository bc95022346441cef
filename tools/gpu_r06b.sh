#!/bin/bash
# round 6, second GPU pass: fused BatchNorm statistics (conv epilogues), tuned BatchNorm streaming kernels, cheaper range audit
TAG=${1:-r06b}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_train_ops_gpu.py tests/test_generator_train_gpu.py tests/test_encoder_train_gpu.py tests/test_cycle_gpu.py tests/test_generator_gpu.py tests/test_conv_gpu.py tests/test_encoder_gpu.py tests/test_render_gpu.py -m gpu -q -x > $OUT/${TAG}_pytest_part.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest_part.log; tail -25 $OUT/${TAG}_pytest_part.log | cut -c1-250
timeout 200 python tools/range_trip_probe.py > $OUT/${TAG}_range_probe.txt 2>&1; tail -20 $OUT/${TAG}_range_probe.txt | cut -c1-300
for wl in train64 infer256; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_$wl.json 2> $OUT/${TAG}_bench_$wl.err; python tools/bench_summary.py $OUT/${TAG}_bench_$wl.json 16
done
timeout 300 python bench.py --workload train64 --train-arith f16x1 --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64_f16x1.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_f16x1.json 3
SMIRK_BN_STATS_UNFUSED=1 timeout 300 python bench.py --workload train64 --steps 10 --warmup 3 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_unfused.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_unfused.json 1
timeout 400 python bench.py --steps 10 --warmup 3 --traffic off --cpu-faces 0 --no-also > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err; python tools/bench_summary.py $OUT/${TAG}_bench_full.json 14
