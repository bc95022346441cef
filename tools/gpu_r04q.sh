#!/bin/bash
# With one hardware queue per stream (GPU_MAX_HW_QUEUES=16, now bench.py's default): re-tune generator streams / chains, and try 24 / 32 queues.
cd /root/repo
b() { tag=$1; shift; python bench.py "$@" --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04q_$tag.json; echo "$tag $(python tools/bench_summary.py gpurun_out/r04q_$tag.json 0 | head -1)"; }
for gb in 128 256; do
  for gs in 1 2 3; do for sc in 0 2; do
    SMIRK_GEN_SPLIT_CHAINS=$sc b b${gb}_gs${gs}_sc${sc} --workload full --global-batch $gb --force-collective --steps 30 --warmup 5 --generator-streams $gs
  done; done
done
for gs in 1 2; do for sc in 0 2; do
  SMIRK_GEN_SPLIT_CHAINS=$sc b b1024_gs${gs}_sc${sc} --workload full --steps 12 --warmup 3 --generator-streams $gs
done; done
b b512_default --workload full --global-batch 512 --force-collective --steps 16 --warmup 4
for q in 24 32; do
  GPU_MAX_HW_QUEUES=$q b b128_q$q --workload full --global-batch 128 --force-collective --steps 30 --warmup 5
  GPU_MAX_HW_QUEUES=$q b train64_q$q --workload train64 --steps 10 --warmup 3
done
b b128_default_again --workload full --global-batch 128 --force-collective --steps 30 --warmup 5
