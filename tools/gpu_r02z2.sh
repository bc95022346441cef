#!/bin/bash
TAG=${1:-r02z2}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
for mb in 256 384 512 683 1024; do
  timeout 300 python bench.py --workload full --steps 4 --warmup 1 --traffic off --cpu-faces 0 --no-roofline --micro-batch $mb > $OUT/${TAG}_bench_full_mb$mb.json 2> $OUT/${TAG}_bench_full_mb$mb.err
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_full_mb$mb.json")); print("full mb=$mb", round(j["value"],1), round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],2))
except Exception as e: print("mb=$mb failed", e)
PY
done
for b in 128 256 512; do
  timeout 300 python bench.py --workload full --steps 6 --warmup 2 --traffic off --cpu-faces 0 --no-roofline --global-batch $b > $OUT/${TAG}_bench_full_gb$b.json 2> $OUT/${TAG}_bench_full_gb$b.err
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_full_gb$b.json")); print("full global-batch=$b (what one rank of a 1024-frame job sees at N=${b})", round(j["value"],1), round(j["ms_per_step"],2), j["config"]["micro_batch"])
except Exception as e: print("gb=$b failed", e)
PY
done
