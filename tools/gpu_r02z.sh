#!/bin/bash
# round-2 call: micro-batch sweep of the full workload (1024 frames, 1 GPU) + train64 after the stage-2 unroll
TAG=${1:-r02z}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_train_ops_gpu.py -q -x > $OUT/${TAG}_pytest_ops.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_ops.log; tail -3 $OUT/${TAG}_pytest_ops.log
timeout 300 python bench.py --workload train64 --steps 8 --warmup 2 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_bench_train64.err
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_train64.json")); r=j["roofline"]
print("train64", j["value"], j["ms_per_step"], j["host_enqueue_ms_per_step"])
for k,v in list(r["kernels"].items())[:14]: print("  ",k,v)
PY
for mb in 167 334 342 501 512; do
  timeout 300 python bench.py --workload full --steps 4 --warmup 1 --traffic off --cpu-faces 0 --no-roofline --micro-batch $mb > $OUT/${TAG}_bench_full_mb$mb.json 2> $OUT/${TAG}_bench_full_mb$mb.err
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_full_mb$mb.json")); print("full mb=$mb", round(j["value"],1), round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],2))
except Exception as e: print("mb=$mb failed", e)
PY
done
