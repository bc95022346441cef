#!/bin/bash
TAG=${1:-r02aa}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_scale_gpu.py tests/test_generator_gpu.py tests/test_conv_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest.log; tail -5 $OUT/${TAG}_pytest.log | cut -c1-250
for i in 1 2; do
timeout 300 python bench.py --workload full --steps 6 --warmup 2 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_full_$i.json 2> $OUT/${TAG}_bench_full_$i.err
python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_full_$i.json")); r=j["roofline"]
print("full", round(j["value"],1), round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],2), j["config"]["micro_batch"])
print(r["kernel"], round(r["achieved"],1), round(r["frac"],4), r["launches_per_pass"], round(r["avg_launch_ms"],4))
for k,v in list(r["kernels"].items())[:12]: print("  ",k,v)
PY
done
