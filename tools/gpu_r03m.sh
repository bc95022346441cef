#!/bin/bash
TAG=${1:-r03m}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
SMIRK_ENCODER_SERIAL=1 timeout 600 python bench.py --workload infer256 --global-batch 1024 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_infer1024_serial.json 2> $OUT/${TAG}_err.txt; echo "rc=$?"
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_infer1024_serial.json")); r=j["roofline"]
print("infer (B=1024, encoder on ONE stream)", round(j["value"],1), round(j["ms_per_step"],2), "profiled kernel ms", round(r["profiled_kernel_ms_per_pass"],2))
for k,v in list(r.get("kernels",{}).items())[:28]: print("  ",k,v)
PY
timeout 600 python bench.py --workload infer256 --global-batch 1024 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_infer1024.json 2>> $OUT/${TAG}_err.txt
python -c "
import json; j=json.load(open('$OUT/${TAG}_bench_infer1024.json')); print('infer (B=1024, three streams)', round(j['value'],1), round(j['ms_per_step'],2))"
