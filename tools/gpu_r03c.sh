#!/bin/bash
# round-3 call C: halo kernel with the early hand-over barrier (SMIRK_HALO_EB) and the prefetched epilogue — parity, EB sweep, timeline, L2 counters
TAG=${1:-r03c}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x > $OUT/${TAG}_pytest_conv.log 2>&1; echo "conv rc=$?"; tail -3 $OUT/${TAG}_pytest_conv.log | cut -c1-300
for EB in 0 4 8; do
  SMIRK_HALO_EB=$EB SMIRK_IGEMM_HALO=all timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k halo > $OUT/${TAG}_pytest_conv_eb$EB.log 2>&1; echo "conv EB=$EB rc=$?"
  echo "== SMIRK_HALO_EB=$EB (B=1024)" >> $OUT/${TAG}_eb_sweep.txt
  SMIRK_IGEMM_HALO=all SMIRK_HALO_EB=$EB timeout 200 python tools/conv_sweep.py --batch 1024 --iters 5 2>&1 | grep -E "enc3|dec3|enc4|dec4|bott|res" >> $OUT/${TAG}_eb_sweep.txt
done
echo "== SMIRK_HALO_EB=default (B=128)" >> $OUT/${TAG}_eb_sweep.txt
SMIRK_IGEMM_HALO=all timeout 200 python tools/conv_sweep.py --batch 128 --iters 10 --ab-env SMIRK_IGEMM_HALO=0 2>&1 | grep -E "enc3|dec3|enc4|dec4|bott|res" >> $OUT/${TAG}_eb_sweep.txt
cut -c1-190 $OUT/${TAG}_eb_sweep.txt
for EB in 0 4; do
  for cfg in "14 512 512 1024 1" "28 256 256 1024 0"; do
    SMIRK_HIP_LIBRARY=/root/repo/smirk_amd/lib_fz/libsmirk_hip_variant.so SMIRK_IGEMM_HALO=all SMIRK_HALO_EB=$EB timeout 200 python tools/halo_timeline.py $cfg 2>&1 | grep -v amdgpu.ids | head -5 >> $OUT/${TAG}_timeline.txt
  done
done
cut -c1-250 $OUT/${TAG}_timeline.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum; do
  rm -rf /tmp/pmc
  SMIRK_IGEMM_HALO=all timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc -o p -- python /root/repo/tools/conv_sweep.py --batch 1024 --only "res" --iters 3 > /tmp/pmc.log 2>&1
  db=$(find /tmp/pmc -name "*.db" | head -1)
  echo "== $c (res 14x14 512->512 reflect, B=1024)" >> $OUT/${TAG}_pmc_halo.txt
  python /root/repo/tools/pmc_summary.py $db $c 2>&1 | head -3 >> $OUT/${TAG}_pmc_halo.txt
done
cat $OUT/${TAG}_pmc_halo.txt | cut -c1-200
