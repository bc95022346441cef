#!/bin/bash
TAG=${1:-r03g}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 300 python tools/mbi_debug.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_mbi_debug.txt; cat $OUT/${TAG}_mbi_debug.txt | cut -c1-300
