#!/bin/bash
# round-2 call: HIP-graph replay of the training CNNs (test + train64 bench with / without graphs)
TAG=${1:-r02x}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 420 python -m pytest tests/test_cycle_gpu.py -q -x -k graphed > $OUT/${TAG}_pytest_graph.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_graph.log
tail -25 $OUT/${TAG}_pytest_graph.log
for mode in graphs nographs; do
  extra=""; [ $mode = graphs ] && extra="--train-graphs"
  timeout 420 python bench.py --workload train64 --steps 8 --warmup 2 --traffic off $extra > $OUT/${TAG}_bench_train64_$mode.json 2> $OUT/${TAG}_bench_train64_$mode.err
  echo "bench train64 $mode rc=$?"; tail -3 $OUT/${TAG}_bench_train64_$mode.err | cut -c1-300
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_train64_$mode.json")); r=j["roofline"]
    print("$mode", j["value"], j["ms_per_step"], j["host_enqueue_ms_per_step"], j["output_stats"])
except Exception as e: print("no line", e)
PY
done
