#!/bin/bash
# round 5, session b: the z-sorted rasteriser and the spill-free encoder kernels against the parity tests; config 3 / config 5 standalone beside the "also" entries
TAG=${1:-r05b}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_scale_gpu.py tests/test_encoder_gpu.py tests/test_dropin_gpu.py tests/test_masking_gpu.py tests/test_chain_gpu.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log
timeout 300 python bench.py --workload infer256 --steps 20 --warmup 5 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_infer256.json 2> $OUT/${TAG}_bench_infer256.err; python tools/bench_summary.py $OUT/${TAG}_bench_infer256.json 12
timeout 300 python bench.py --workload train64 --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_bench_train64.err; python tools/bench_summary.py $OUT/${TAG}_bench_train64.json 6
timeout 300 python bench.py --workload train64 --train-arith f16x1 --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64_f16x1.json 2> $OUT/${TAG}_bench_train64_f16x1.err; python tools/bench_summary.py $OUT/${TAG}_bench_train64_f16x1.json 4
timeout 400 python bench.py --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_full_also.json 2> $OUT/${TAG}_bench_full_also.err; python tools/bench_summary.py $OUT/${TAG}_bench_full_also.json 12
