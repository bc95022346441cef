#!/bin/bash
# Round 4, session c: the 128-frame shard with two generator streams (partial rounds of one pass's deep layers overlap the next pass's), halo kernel on every eligible layer
TAG=${1:-r04c}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
run() { # name, env..., -- args
  name=$1; shift
  env "$@" timeout 300 python bench.py --workload full --force-collective --steps 30 --warmup 5 --cpu-faces 0 --traffic off --no-roofline $ARGS 2>/dev/null | grep '^{' > $OUT/${TAG}_$name.json
  python -c "
import json; j=json.load(open('$OUT/${TAG}_$name.json')); print('$name', round(j['value'],1), 'faces/s', round(j['ms_per_step'],2), 'ms')"
}
ARGS="--global-batch 128" run b128_base A=1
ARGS="--global-batch 128 --generator-streams 2" run b128_gs2 A=1
ARGS="--global-batch 128 --generator-streams 3" run b128_gs3 A=1
ARGS="--global-batch 128" run b128_haloall SMIRK_IGEMM_HALO=all
ARGS="--global-batch 128 --generator-streams 2" run b128_gs2_haloall SMIRK_IGEMM_HALO=all
ARGS="--global-batch 256 --generator-streams 2" run b256_gs2 A=1
ARGS="--global-batch 1024 --generator-streams 2" run b1024_gs2 A=1
ARGS="--global-batch 128 --micro-batch 64 --generator-streams 2" run b128_mb64_gs2 A=1
