#!/bin/bash
OUT=/root/repo/gpurun_out
cd /root/repo
timeout 600 python -m pytest tests/test_scale_gpu.py tests/test_encoder_gpu.py tests/test_masking_gpu.py -x -q 2>&1 | tail -3
run() { tag=$1; shift
  timeout 600 python bench.py --workload full --steps 4 --warmup 1 --cpu-faces 0 --traffic off "$@" > $OUT/r02x_$tag.json 2>$OUT/r02x_$tag.err
  python - <<PY
import json
j=json.load(open("$OUT/r02x_$tag.json")); r=j["roofline"]
print("$tag", round(j["value"],1), "faces/s", round(j["ms_per_step"],2), "ms/step; host", round(j["host_enqueue_ms_per_step"],1), "; dominant", r["kernel"], round(r["achieved"],1), r["unit"], round(r["frac"],4))
PY
}
run mb128_g1 --micro-batch 128 --generator-streams 1
run mb128_g2 --micro-batch 128 --generator-streams 2
run mb167_g1 --micro-batch 167 --generator-streams 1
run mb167_g2 --micro-batch 167 --generator-streams 2
run mb128_g3 --micro-batch 128 --generator-streams 3
run mb84_g2 --micro-batch 84 --generator-streams 2
