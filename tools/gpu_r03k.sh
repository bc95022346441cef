#!/bin/bash
# round-3 call K: chain test, eval-mode generator backward, raster share on realistic triangles, kernel stats
TAG=${1:-r03k}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_generator_train_gpu.py -q -x -s > $OUT/${TAG}_pytest_chain.log 2>&1; echo "chain/eval pytest rc=$?"; grep -E "passed|failed|step-1 chain|eval-mode generator|Error|assert" $OUT/${TAG}_pytest_chain.log | cut -c1-1200 | head -20
for basis in smooth random; do
  timeout 600 python bench.py --flame-basis $basis --traffic off --cpu-faces 0 --steps 3 > $OUT/${TAG}_bench_full_$basis.json 2> $OUT/${TAG}_bench_full_$basis.err; echo "bench $basis rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_full_$basis.json")); r=j["roofline"]
    k=r["kernels"]
    print("$basis", round(j["value"],1), round(j["ms_per_step"],2), "raster_tile", k.get("raster_tile"), "raster_face_setup", k.get("raster_face_setup"))
except Exception as e: print("no line", e)
PY
done
