#!/bin/bash
# RotatingPipeline (infer256 lanes) parity + A/B; the new defaults (16 hardware queues, 2 generator streams) on the shard sizes.
cd /root/repo
python -m pytest tests/test_scale_gpu.py -q -x -k "rotating or repeated" 2>&1 | tail -2
b() { tag=$1; shift; python bench.py "$@" --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04r_$tag.json; echo "$tag $(python tools/bench_summary.py gpurun_out/r04r_$tag.json 0 | head -1)"; }
for lanes in 1 2 3; do b infer256_lanes$lanes --workload infer256 --steps 60 --warmup 10 --infer-lanes $lanes; done
for gb in 128 256 512 1024; do b shard$gb --workload full --global-batch $gb --force-collective --steps $((3840 / gb > 30 ? 30 : 3840 / gb + 8)) --warmup 4; done
