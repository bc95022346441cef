#!/bin/bash
TAG=${1:-r03r}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_raster_differential.py tests/test_dropin_gpu.py tests/test_generator_train_gpu.py tests/test_cycle_gpu.py tests/test_chain_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log | cut -c1-300
timeout 600 python bench.py --traffic off --cpu-faces 0 --steps 3 > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_err.txt
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_full.json")); k=j["roofline"]["kernels"]
print("full", round(j["value"],1), round(j["ms_per_step"],2), "raster_tile", k.get("raster_tile"))
PY
timeout 900 python bench.py --workload train64 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64.json 2>> $OUT/${TAG}_err.txt
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_train64.json")); k=j["roofline"]["kernels"]
print("train64", round(j["value"],1), round(j["ms_per_step"],2), "host", round(j.get("host_enqueue_ms_per_step",0),1), {n:v for n,v in k.items() if "pack" in n})
PY
