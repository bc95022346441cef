#!/bin/bash
# round 6, pre-final: the whole GPU suite + smoke on the candidate sources, quick lines of configs 3 / 4
TAG=${1:-r06h}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -12 $OUT/${TAG}_pytest.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/${TAG}_smoke.txt
timeout 300 python bench.py --workload infer256 --steps 20 --warmup 5 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_infer256.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_bench_infer256.json 6
timeout 400 python bench.py --steps 10 --warmup 3 --traffic off --cpu-faces 0 --no-also > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err; python tools/bench_summary.py $OUT/${TAG}_bench_full.json 14
