#!/bin/bash
# round-3 call E: image-resident MBConv + faster encoder head — full GPU suite, encoder / full benches
TAG=${1:-r03e}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_encoder_gpu.py -q -x > $OUT/${TAG}_pytest_enc.log 2>&1; echo "encoder pytest rc=$?"; tail -6 $OUT/${TAG}_pytest_enc.log | cut -c1-400
for wl in infer256 full; do
  timeout 600 python bench.py --workload $wl --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_$wl.json 2> $OUT/${TAG}_bench_$wl.err; echo "bench $wl rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_$wl.json")); r=j["roofline"]
    print("$wl", round(j["value"],1), round(j["ms_per_step"],2), r.get("kernel"), round(r.get("frac"),4))
    for k,v in list(r.get("kernels",{}).items())[:24]: print("  ",k,v)
except Exception as e: print("no line", e)
PY
done
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_all.log 2>&1; echo "all gpu pytest rc=$?"; tail -5 $OUT/${TAG}_pytest_all.log | cut -c1-400
