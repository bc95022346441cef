#!/bin/bash
# Kernel-selection defaults decided while the backbone streams were queued behind generator kernels: re-measured with one hardware queue per stream.
cd /root/repo
b() { tag=$1; shift; python bench.py "$@" --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04x_$tag.json; echo "$tag $(python tools/bench_summary.py gpurun_out/r04x_$tag.json 0 | head -1)"; }
S128="--workload full --global-batch 128 --force-collective --steps 30 --warmup 4"
S1024="--workload full --global-batch 1024 --force-collective --steps 12 --warmup 3"
b base_128 $S128
SMIRK_DISABLE_MBCONV_IMAGE=1 b no_mbconv_image_128 $S128
SMIRK_DISABLE_ENCODER_HEAD_FUSED=1 b no_head_fused_128 $S128
SMIRK_IGEMM_HALO=all b halo_all_128 $S128
SMIRK_HALO_EB=4 b halo_eb4_128 $S128
SMIRK_MBCONV_FUSE_DS=0 b no_fuse_ds_128 $S128
b base_1024 $S1024
SMIRK_DISABLE_MBCONV_IMAGE=1 b no_mbconv_image_1024 $S1024
SMIRK_HALO_EB=4 b halo_eb4_1024 $S1024
SMIRK_IGEMM_HALO=all b halo_all_1024 $S1024
b base_128_again $S128
