#!/bin/bash
# round 5, session c: parameter-layout weight gradients, conv_halo f16x1, batched bin loop of the rasteriser; the driver's own command with child-process also-entries
TAG=${1:-r05c}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_generator_train_gpu.py tests/test_cycle_gpu.py tests/test_train_scale_gpu.py tests/test_render_gpu.py tests/test_conv_gpu.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log
timeout 200 python -m pytest tests/test_scale_gpu.py -m gpu -q -x -k "flame_renderer or full_pipeline" > $OUT/${TAG}_pytest2.log 2>&1; tail -2 $OUT/${TAG}_pytest2.log
( time timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err ) 2>&1 | grep real; python tools/bench_summary.py $OUT/${TAG}_bench_default.json 14; tail -2 $OUT/${TAG}_bench_default.err
