#!/bin/bash
# diagnostic session: concurrency diff, full GPU suite (no -x), bench full
TAG=${1:-r02b}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python tools/overlap_diff.py 128 > $OUT/${TAG}_overlap_diff.txt 2>&1; echo "overlap_diff rc=$?"; cat $OUT/${TAG}_overlap_diff.txt | tail -12
timeout 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
tail -15 $OUT/${TAG}_pytest.log
timeout 900 python bench.py --workload full --steps 5 --warmup 2 > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err
echo "bench full rc=$?"; head -c 1500 $OUT/${TAG}_bench_full.json; echo; tail -5 $OUT/${TAG}_bench_full.err
