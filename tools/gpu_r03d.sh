#!/bin/bash
# round-3 call D: fused encoder head — parity + encoder / full benches
TAG=${1:-r03d}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_conv_gpu.py tests/test_generator_train_gpu.py tests/test_cycle_gpu.py tests/test_train_ops_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.log | cut -c1-300
for wl in infer256 full; do
  timeout 600 python bench.py --workload $wl --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_$wl.json 2> $OUT/${TAG}_bench_$wl.err; echo "bench $wl rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_$wl.json")); r=j["roofline"]
    print("$wl", round(j["value"],1), round(j["ms_per_step"],2), r.get("kernel"), round(r.get("frac"),4))
    for k,v in list(r.get("kernels",{}).items())[:22]: print("  ",k,v)
except Exception as e: print("no line", e)
PY
done
SMIRK_DISABLE_ENCODER_HEAD_FUSED=1 timeout 600 python bench.py --workload infer256 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_infer256_nohead.json 2>/dev/null; python -c "
import json; j=json.load(open('$OUT/${TAG}_bench_infer256_nohead.json')); print('infer256 without fused head', round(j['value'],1), round(j['ms_per_step'],2))"
