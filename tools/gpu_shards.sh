#!/bin/bash
# per-rank shard regime of the 8 / 4 / 2 / 1-GPU job on ONE MI355X with the RCCL all-gather really enqueued (world-size-1 group)
TAG=${1:-shards}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
for gb in 1024 128 256 512; do
  timeout 300 python bench.py --workload full --global-batch $gb --force-collective --steps 20 --warmup 5 --cpu-faces 0 --traffic off 2> $OUT/${TAG}_shard_${gb}.err | grep '^{' > $OUT/${TAG}_shard_${gb}.json
  python tools/bench_summary.py $OUT/${TAG}_shard_${gb}.json 0 | head -1
done
