#!/bin/bash
OUT=/root/repo/gpurun_out
cd /root/repo
run() { # tag env... 
  tag=$1; shift
  env "$@" timeout 600 python bench.py --workload full --steps 4 --warmup 1 --cpu-faces 0 --traffic off $EXTRA > $OUT/r02w_$tag.json 2>$OUT/r02w_$tag.err
  python - <<PY
import json
j=json.load(open("$OUT/r02w_$tag.json")); r=j["roofline"]
print("$tag", round(j["value"],1), "faces/s", round(j["ms_per_step"],2), "ms/step; dominant", r["kernel"], round(r["achieved"],1), "TFLOP/s", round(r["avg_launch_ms"],4), "ms")
PY
}
EXTRA="--micro-batch 128" run mb128_default SMIRK_X=1
EXTRA="--micro-batch 128" run mb128_ppall SMIRK_IGEMM_PP=all
EXTRA="--micro-batch 167" run mb167_default SMIRK_X=1
EXTRA="--micro-batch 167" run mb167_ppall SMIRK_IGEMM_PP=all
EXTRA="--micro-batch 167" run mb167_pp0 SMIRK_IGEMM_PP=0
EXTRA="--micro-batch 128" run mb128_pp0 SMIRK_IGEMM_PP=0
