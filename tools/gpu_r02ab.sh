#!/bin/bash
TAG=${1:-r02ab}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_scale_gpu.py -q -x -k "encoder or pipeline" > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log | cut -c1-250
for i in 1 2; do
timeout 300 python bench.py --workload full --steps 6 --warmup 2 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_full_$i.json 2> $OUT/${TAG}_bench_full_$i.err
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_full_$i.json")); r=j["roofline"]
print("full", round(j["value"],1), round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],2), j["config"]["micro_batch"])
for k,v in list(r["kernels"].items())[:28]:
    if "dwconv" in k or "stem" in k or "maxpool" in k or "pack" in k: print("  ",k,v)
PY
done
timeout 300 python bench.py --workload infer256 --steps 10 --warmup 3 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_infer256.json 2> $OUT/${TAG}_bench_infer256.err
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_infer256.json")); print("infer256", round(j["value"],1), round(j["ms_per_step"],3))
PY
