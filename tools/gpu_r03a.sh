#!/bin/bash
# round-3 call A: halo-staged conv kernel (parity + per-layer A/B), the new parity tests, a first bench line
TAG=${1:-r03a}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x > $OUT/${TAG}_pytest_conv.log 2>&1; echo "conv rc=$?"; tail -5 $OUT/${TAG}_pytest_conv.log | cut -c1-300
for B in 128 1024; do
  SMIRK_IGEMM_HALO=all timeout 300 python tools/conv_sweep.py --batch $B --iters 5 --only "" --ab-env SMIRK_IGEMM_HALO=0 2>&1 | grep -E "enc3|dec3|enc4|dec4|bott|res|total" > $OUT/${TAG}_sweep_B$B.txt; cat $OUT/${TAG}_sweep_B$B.txt | cut -c1-200
done
timeout 900 python -m pytest tests/test_scale_gpu.py tests/test_masking_gpu.py tests/test_rccl_gpu.py tests/test_train_scale_gpu.py -q -x > $OUT/${TAG}_pytest_new.log 2>&1; echo "new rc=$?"; tail -8 $OUT/${TAG}_pytest_new.log | cut -c1-400
timeout 600 python bench.py --traffic off > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err; echo "bench rc=$?"; python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_full.json")); r=j["roofline"]
    print(j["value"], j["ms_per_step"], r.get("kernel"), r.get("frac"))
    for k,v in list(r.get("kernels",{}).items())[:14]: print("  ",k,v)
except Exception as e: print("no line", e)
PY
