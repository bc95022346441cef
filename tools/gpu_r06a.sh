#!/bin/bash
# round 6, first GPU pass: the suite with the always-on range flag + new tests, smoke, and today's baseline lines of configs 3 / 4 / 5
TAG=${1:-r06a}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -30 $OUT/${TAG}_pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/${TAG}_smoke.txt
for wl in train64 infer256; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_$wl.json 2> $OUT/${TAG}_bench_$wl.err; python tools/bench_summary.py $OUT/${TAG}_bench_$wl.json 12
done
timeout 300 python bench.py --workload train64 --train-arith f16x1 --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64_f16x1.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_train64_f16x1.json 3
timeout 400 python bench.py --steps 10 --warmup 3 --traffic off --cpu-faces 0 --no-also > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err; python tools/bench_summary.py $OUT/${TAG}_bench_full.json 12
