#!/bin/bash
cd /root/repo
OUT=/root/repo/gpurun_out; mkdir -p $OUT
for wg in 512 768 1024 1536 2048; do echo "== SMIRK_WGRAD_SPLIT_WG=$wg"; SMIRK_WGRAD_SPLIT_WG=$wg python tools/wgrad_sweep.py 64 2>&1 | grep -v amdgpu.ids | egrep "H= (56|28|14)|k=1|total"; done | tee $OUT/r02aj_wgrad_split_sweep.txt
