#!/bin/bash
# round-2 call: one-launch (cooperative) BatchNorm — unit tests, whole-network training tests, train64 A/B
TAG=${1:-r02ah}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_train_ops_gpu.py -q -x -k "batchnorm" > $OUT/${TAG}_pytest_bn.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_bn.log; tail -4 $OUT/${TAG}_pytest_bn.log | cut -c1-200
timeout 600 python -m pytest tests/test_generator_train_gpu.py tests/test_encoder_train_gpu.py tests/test_cycle_gpu.py -q -x > $OUT/${TAG}_pytest_train.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_train.log; tail -3 $OUT/${TAG}_pytest_train.log | cut -c1-200
for mode in 1 0; do
  SMIRK_BN_FUSED=$mode timeout 300 python bench.py --workload train64 --steps 8 --warmup 2 --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_train64_fused$mode.json 2> $OUT/${TAG}_bench_train64_fused$mode.err
  echo "bench rc=$?"; grep -v "amdgpu.ids\|Warning\|run_backward" $OUT/${TAG}_bench_train64_fused$mode.err | tail -2 | cut -c1-200
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_train64_fused$mode.json")); r=j["roofline"]
    print("SMIRK_BN_FUSED=$mode", round(j["value"],1), round(j["ms_per_step"],2), "host", round(j["host_enqueue_ms_per_step"],2), "timeouts", j.get("bn_fused_barrier_timeouts"))
    for k,v in list(r["kernels"].items())[:30]:
        if k.startswith(("bn_","colsum")): print("   ",k,v)
except Exception as e: print("no line", e)
PY
done
