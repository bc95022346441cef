#!/bin/bash
TAG=${1:-r02c}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python tools/overlap_diff2.py 128 > $OUT/${TAG}_overlap_diff2.txt 2>&1; echo "overlap_diff2 rc=$?"; grep -v amdgpu.ids $OUT/${TAG}_overlap_diff2.txt | tail -30
timeout 900 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
tail -8 $OUT/${TAG}_pytest.log
for mb in 166 83; do
  timeout 600 python bench.py --workload full --micro-batch $mb --global-batch $((mb*6)) --steps 4 --warmup 1 --cpu-faces 0 --traffic off > $OUT/${TAG}_bench_mb$mb.json 2>$OUT/${TAG}_bench_mb$mb.err
  echo "bench mb=$mb rc=$?"; python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_mb$mb.json")); r=j["roofline"]
print(j["value"], j["ms_per_step"], j["host_enqueue_ms_per_step"], r["kernel"], r["achieved"], r["avg_launch_ms"])
for k,v in list(r["kernels"].items())[:8]: print("  ",k,v)
PY
done
