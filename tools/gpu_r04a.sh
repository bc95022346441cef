#!/bin/bash
# First GPU session planned for round 4 (written at the end of round 3, when the budget was spent): everything here is a measurement that round 3 still owes.
#   build first (in the build container):  bash tools/build_variant.sh "-DMB_ES2=48" mbconv.hip
TAG=${1:-r04a}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
# 1. the whole GPU suite on the shipped library (the LDS-layout changes of round 3 were validated file by file, not as a suite)
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -2 $OUT/${TAG}_pytest.log
# 2. same-box A/B: stride-2 fused MBConv with the conflict-free E stride (48 floats) against the shipped 36
for lib in "" /root/repo/smirk_amd/lib_fz/libsmirk_hip_variant.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  SMIRK_HIP_LIBRARY=$lib timeout 120 python bench.py --workload infer256 --steps 10 --warmup 3 --cpu-faces 0 --traffic off > $OUT/${TAG}_infer256_${lib:+es48}.json 2>/dev/null
  python -c "
import json; j=json.load(open('$OUT/${TAG}_infer256_${lib:+es48}.json')); k=j['roofline']['kernels']
print('lib=${lib:-shipped}', round(j['value'],1), {n:v['ms_per_pass'] for n,v in k.items() if 'mbconv_fused' in n})"
done
# 3. counter census + clocks of the training step and of config 3 (planning data for BatchNorm fusion / the encoder)
for wl in train64 infer256; do
  timeout 300 python tools/pmc_census.py $wl $OUT/${TAG}_pmc_census_${wl}.txt > /dev/null 2>&1
  timeout 120 python tools/pmc_clock.py $wl $OUT/${TAG}_pmc_clock_${wl}.txt > /dev/null 2>&1
done
ls -la $OUT | grep ${TAG}
