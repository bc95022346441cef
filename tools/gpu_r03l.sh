#!/bin/bash
TAG=${1:-r03l}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_masking_gpu.py tests/test_scale_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log | cut -c1-300
for wl in infer256 full; do
  timeout 600 python bench.py --workload $wl --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_$wl.json 2> $OUT/${TAG}_bench_$wl.err; echo "bench $wl rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_$wl.json")); r=j["roofline"]
    print("$wl", round(j["value"],1), round(j["ms_per_step"],2), r.get("kernel"), round(r.get("frac"),4))
    for k,v in list(r.get("kernels",{}).items())[:26]:
        if any(t in k for t in ("mbconv","encoder_head","maxpool_sq","conv_halo","dwconv","igemm")): print("  ",k,v)
except Exception as e: print("no line", e)
PY
done
