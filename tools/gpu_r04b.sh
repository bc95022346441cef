#!/bin/bash
# Round 4, session b: the per-rank shard regime of the 8 / 4 / 2-GPU job measured on ONE MI355X with RCCL really enqueued (verdict item 2), plus the
# per-layer baseline of the 224^2 / 112^2 generator layers at B = 128 and B = 1024 (verdict item 1) on the round-3 kernels.
TAG=${1:-r04b}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
for gb in 1024 128 256 512; do
  timeout 300 python bench.py --workload full --global-batch $gb --force-collective --steps 20 --warmup 5 --cpu-faces 0 --traffic off > $OUT/${TAG}_shard_${gb}.json 2> $OUT/${TAG}_shard_${gb}.err
  python - <<PY
import json
j=json.load(open('$OUT/${TAG}_shard_${gb}.json'))
k=j['roofline']['kernels']
print('gb=$gb', round(j['value'],1), 'faces/s', round(j['ms_per_step'],2), 'ms', j['config']['collective'][:40], 'host', round(j['host_enqueue_ms_per_step'],2), 'kernel ms', round(j['roofline']['profiled_kernel_ms_per_pass'],2))
PY
done
for B in 128 1024; do
  echo "== conv_sweep B=$B" ; timeout 300 python tools/conv_sweep.py --batch $B --iters 5 2>&1 | grep -v amdgpu
done > $OUT/${TAG}_conv_sweep.txt
cat $OUT/${TAG}_conv_sweep.txt
