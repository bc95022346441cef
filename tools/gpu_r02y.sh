#!/bin/bash
# round-2 call: TRAIN-mode encoder on three streams (test + train64 bench A/B)
TAG=${1:-r02y}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_encoder_train_gpu.py tests/test_cycle_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log
tail -12 $OUT/${TAG}_pytest.log | cut -c1-300
for mode in streams serial; do
  [ $mode = serial ] && export SMIRK_ENCODER_TRAIN_SERIAL=1
  timeout 420 python bench.py --workload train64 --steps 8 --warmup 2 --traffic off > $OUT/${TAG}_bench_train64_$mode.json 2> $OUT/${TAG}_bench_train64_$mode.err
  echo "bench train64 $mode rc=$?"; grep -v "amdgpu.ids\|Warning\|run_backward" $OUT/${TAG}_bench_train64_$mode.err | tail -3 | cut -c1-300
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_train64_$mode.json")); r=j["roofline"]
    print("$mode", j["value"], j["ms_per_step"], j["host_enqueue_ms_per_step"], j["output_stats"])
except Exception as e: print("no line", e)
PY
done
