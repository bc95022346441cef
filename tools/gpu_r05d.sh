#!/bin/bash
# round 5, session d: halo-tiled / K-padded mbconv_image variants against float64 and the kernels they replace; config 3 with and without them (same box)
TAG=${1:-r05d}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -6 $OUT/${TAG}_pytest.log
timeout 400 python -m pytest tests/test_scale_gpu.py -m gpu -q -x -k "encoder" > $OUT/${TAG}_pytest2.log 2>&1; tail -2 $OUT/${TAG}_pytest2.log
for v in tile notile tile notile; do
  if [ $v = notile ]; then export SMIRK_DISABLE_MBCONV_TILE=1; else unset SMIRK_DISABLE_MBCONV_TILE; fi
  timeout 300 python bench.py --workload infer256 --steps 30 --warmup 5 --traffic off --cpu-faces 0 > $OUT/${TAG}_infer256_$v.json 2> $OUT/${TAG}_infer256_$v.err; echo "== $v"; python tools/bench_summary.py $OUT/${TAG}_infer256_$v.json 9
done
unset SMIRK_DISABLE_MBCONV_TILE
