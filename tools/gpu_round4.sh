#!/bin/bash
TAG=${1:-r02s}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
python tools/overlap_diff4.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -3 | tee $OUT/${TAG}_diff4.txt
DIFF5_SHORT=1 python tools/overlap_diff5.py 1 2>&1 | grep aggressor | tee $OUT/${TAG}_diff5.txt
timeout 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
tail -6 $OUT/${TAG}_pytest.log
timeout 900 python bench.py --workload full --steps 5 --warmup 2 > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err
echo "bench full rc=$?"; python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_full.json")); r=j["roofline"]
print(j["value"], j["ms_per_step"], j["host_enqueue_ms_per_step"], j["output_stats"])
print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_ms"], r["traffic"])
for k,v in list(r["kernels"].items())[:14]: print("  ",k,v)
PY
