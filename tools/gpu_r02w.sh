#!/bin/bash
# round-2 call A: LDS transpose-read probe, the split-fp16 weight-gradient kernels (unit tests for the three kernel modes, per-layer sweep), whole-network
# training tests on the new default, a short train64 bench line
TAG=${1:-r02w}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
python tools/tr_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_tr_probe.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_wgrad_modes.txt
import sys, torch
sys.path.insert(0, "tests")
import test_train_ops_gpu as t
T, ops = t._ops()
import torch.nn.functional as F
for (B, H, W, cin, cout, k) in [(2, 8, 8, 32, 64, 3), (2, 16, 16, 32, 32, 3), (2, 6, 6, 256, 512, 1)]:
    g = t._gen(1)
    xs, x64 = t._act(torch.randn(B, H, W, cin, generator=g)); ds, d64 = t._act(torch.randn(B, H, W, cout, generator=g))
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x64, w, padding=k // 2).backward(d64)
    for mode in (0, 1, 2, 17, 18):
        ops.lib.smirk_conv_wgrad_set_mode(mode)
        dw = ops.wgrad(ds, xs, B, H, W, cout, cin, k)
        print((B, H, W, cin, cout, k), "mode", mode, "rel err", float(t._rel(T._to_conv_weight_grad(dw, cout, cin, k).cpu(), w.grad)))
    ops.lib.smirk_conv_wgrad_set_mode(-1)
PY
timeout 600 python -m pytest tests/test_train_ops_gpu.py -q -k "weight_gradient or transpose or final" > $OUT/${TAG}_pytest_wgrad.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_wgrad.log
tail -5 $OUT/${TAG}_pytest_wgrad.log
for m in 0 1 2; do SMIRK_WGRAD_F16=$m python tools/wgrad_sweep.py 64 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_wgrad_sweep_mode$m.txt | tail -24; done
timeout 900 python -m pytest tests/test_generator_train_gpu.py tests/test_encoder_train_gpu.py tests/test_cycle_gpu.py -q > $OUT/${TAG}_pytest_train.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_pytest_train.log
tail -5 $OUT/${TAG}_pytest_train.log
timeout 600 python bench.py --workload train64 --steps 5 --warmup 2 --traffic off > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_bench_train64.err
echo "bench train64 rc=$?"; python - <<PY
import json
j=json.load(open("$OUT/${TAG}_bench_train64.json")); r=j["roofline"]
print(j["value"], j["ms_per_step"], j["host_enqueue_ms_per_step"], j["output_stats"])
print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_ms"])
for k,v in list(r["kernels"].items())[:16]: print("  ",k,v)
PY
