#!/bin/bash
TAG=${1:-r03o}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 900 python bench.py --workload train64 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_err.txt; echo "rc=$?"
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_train64.json")); r=j["roofline"]
print("train64", round(j["value"],1), round(j["ms_per_step"],2), "host", j.get("host_enqueue_ms_per_step"), "profiled kernel ms", round(r["profiled_kernel_ms_per_pass"],2))
for k,v in list(r.get("kernels",{}).items())[:28]: print("  ",k,v)
PY
SMIRK_IGEMM_HALO=0 timeout 900 python bench.py --workload train64 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_nohalo.json 2>> $OUT/${TAG}_err.txt
python -c "
import json; j=json.load(open('$OUT/${TAG}_bench_train64_nohalo.json')); print('train64 without conv_halo', round(j['value'],1), round(j['ms_per_step'],2))"
