#!/bin/bash
# Stream priorities and the encoder's stream count on top of one-queue-per-stream: generator streams high, or the front end high, or the backbones on one stream.
cd /root/repo
b() { tag=$1; shift; python bench.py "$@" --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04w_$tag.json; echo "$tag $(python tools/bench_summary.py gpurun_out/r04w_$tag.json 0 | head -1)"; }
for gb in 128 1024; do
  st=$((gb == 128 ? 30 : 12))
  b base_$gb --workload full --global-batch $gb --force-collective --steps $st --warmup 4
  SMIRK_GEN_STREAM_PRIORITY=-1 b genhigh_$gb --workload full --global-batch $gb --force-collective --steps $st --warmup 4
  SMIRK_FRONT_STREAM_PRIORITY=-1 b fronthigh_$gb --workload full --global-batch $gb --force-collective --steps $st --warmup 4
  SMIRK_ENCODER_SERIAL=1 b encserial_$gb --workload full --global-batch $gb --force-collective --steps $st --warmup 4
done
b mb512 --workload full --micro-batch 512 --steps 12 --warmup 3
b base_again_1024 --workload full --steps 12 --warmup 3
