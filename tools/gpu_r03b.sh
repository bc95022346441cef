#!/bin/bash
# round-3 call B: halo kernel — DMA placement sweep (SMIRK_HALO_NL), phase timeline (debug-hook variant library), bench with the halo kernel on every deep layer
TAG=${1:-r03b}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
for NL in 0 1 2 3; do
  echo "== SMIRK_HALO_NL=$NL (B=1024)" >> $OUT/${TAG}_nl_sweep.txt
  SMIRK_IGEMM_HALO=all SMIRK_HALO_NL=$NL timeout 200 python tools/conv_sweep.py --batch 1024 --iters 5 2>&1 | grep -E "enc3|dec3|enc4|dec4|bott|res" >> $OUT/${TAG}_nl_sweep.txt
done
cut -c1-120 $OUT/${TAG}_nl_sweep.txt
for NL in 3 0; do
  for cfg in "14 512 512 1024 1" "56 128 128 1024 0"; do
    SMIRK_HIP_LIBRARY=/root/repo/smirk_amd/lib_fz/libsmirk_hip_variant.so SMIRK_IGEMM_HALO=all SMIRK_HALO_NL=$NL timeout 200 python tools/halo_timeline.py $cfg >> $OUT/${TAG}_timeline.txt 2>&1
  done
done
grep -v "amdgpu.ids" $OUT/${TAG}_timeline.txt | cut -c1-220
SMIRK_IGEMM_HALO=all timeout 900 python bench.py > $OUT/${TAG}_bench_full_haloall.json 2> $OUT/${TAG}_bench_full_haloall.err; echo "bench rc=$?"; python - <<PY
import json
try:
    j=json.load(open("$OUT/${TAG}_bench_full_haloall.json")); r=j["roofline"]
    print(j["value"], j["ms_per_step"], r.get("kernel"), r.get("frac"), "traffic", r.get("traffic"), "alg", r.get("algorithmic_bytes_per_launch"))
    for k,v in list(r.get("kernels",{}).items())[:8]: print("  ",k,v)
except Exception as e: print("no line", e)
PY
