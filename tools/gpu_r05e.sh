#!/bin/bash
TAG=${1:-r05e}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_render_gpu.py tests/test_masking_gpu.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; tail -2 $OUT/${TAG}_pytest.log
timeout 300 python -m pytest tests/test_scale_gpu.py -m gpu -q -x -k "flame_renderer or full_pipeline or hull_mask" > $OUT/${TAG}_pytest2.log 2>&1; tail -2 $OUT/${TAG}_pytest2.log
timeout 120 python tools/encoder_chain.py 256 expression > $OUT/${TAG}_chain_large_256.txt 2>&1; cat $OUT/${TAG}_chain_large_256.txt | cut -c1-120
timeout 120 python tools/encoder_chain.py 256 pose > $OUT/${TAG}_chain_small_256.txt 2>&1; tail -1 $OUT/${TAG}_chain_small_256.txt
timeout 300 python bench.py --workload infer256 --steps 30 --warmup 5 --traffic off --cpu-faces 0 > $OUT/${TAG}_infer256.json 2> $OUT/${TAG}_infer256.err; python tools/bench_summary.py $OUT/${TAG}_infer256.json 6
