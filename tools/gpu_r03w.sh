#!/bin/bash
# copyBuffer census: are the ~2.3 k __amd_rocclr_copyBuffer dispatches of the full workload set-up (parameter uploads) or per step?  + infer256 kernel stats
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_a
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/rp_a -o p -- python /root/repo/bench.py --workload full --steps 8 --warmup 1 --no-roofline --cpu-faces 0 > /tmp/rp_a.log 2>&1
db=$(find /tmp/rp_a -name "*.db" | head -1)
python /root/repo/tools/rocprof_summary.py $db $OUT/r03w_kernel_stats_full_steps8.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload full --steps 8 --warmup 1 --no-roofline --cpu-faces 0" | grep -E "total kernel|copyBuffer|conv_halo_kernel<6"
rm -rf /tmp/rp_b
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/rp_b -o p -- python /root/repo/bench.py --workload infer256 --steps 2 --warmup 1 --no-roofline --cpu-faces 0 > /tmp/rp_b.log 2>&1
db=$(find /tmp/rp_b -name "*.db" | head -1)
python /root/repo/tools/rocprof_summary.py $db $OUT/r03w_kernel_stats_infer256.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload infer256 --steps 2 --warmup 1 --no-roofline --cpu-faces 0" | head -12 | cut -c1-120
