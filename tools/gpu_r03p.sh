#!/bin/bash
TAG=${1:-r03p}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_generator_train_gpu.py tests/test_encoder_train_gpu.py tests/test_cycle_gpu.py -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log | cut -c1-300
for v in default 0; do
  if [ $v = 0 ]; then export SMIRK_BN_SMALL=0; fi
  timeout 900 python bench.py --workload train64 --traffic off --cpu-faces 0 --no-roofline > $OUT/${TAG}_bench_train64_$v.json 2>> $OUT/${TAG}_err.txt
  python -c "
import json; j=json.load(open('$OUT/${TAG}_bench_train64_$v.json')); print('train64 SMIRK_BN_SMALL=$v', round(j['value'],1), round(j['ms_per_step'],2), 'host', round(j.get('host_enqueue_ms_per_step',0),1))"
done
