#!/bin/bash
TAG=${1:-r03q}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
for v in 0 512 1024 4096 16384; do
  SMIRK_BN_SMALL=$v timeout 900 python bench.py --workload train64 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_$v.json 2>> $OUT/${TAG}_err.txt
  python -c "
import json; j=json.load(open('$OUT/${TAG}_bench_train64_$v.json')); print('train64 SMIRK_BN_SMALL=$v', round(j['value'],1), round(j['ms_per_step'],2), 'host', round(j.get('host_enqueue_ms_per_step',0),1))" | tee -a $OUT/${TAG}_bn_small_sweep.txt
done
