#!/bin/bash
# Weight gradients on a low-priority second stream (v2: references held until the join, no record_stream), num_batches_tracked inside the BatchNorm launch.
cd /root/repo
python -m pytest tests/test_train_ops_gpu.py tests/test_generator_train_gpu.py tests/test_encoder_train_gpu.py -q -x 2>&1 | tail -3
for cfg in "0 1" "1 1" "1 0" "0 1" "1 1"; do
  set -- $cfg
  SMIRK_TRAIN_WGRAD_STREAM=$1 SMIRK_TRAIN_WGRAD_PRIORITY=$2 python bench.py --workload train64 --steps 12 --warmup 3 --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04u_train64_side$1_prio$2.json
  echo "wgrad side stream=$1 priority class=$2 $(python tools/bench_summary.py gpurun_out/r04u_train64_side$1_prio$2.json 0 | head -1)"
done
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace -d /tmp/rp_t -o p -- python /root/repo/bench.py --workload train64 --steps 2 --warmup 2 --no-roofline --cpu-faces 0 --traffic off > /tmp/rp_t.log 2>&1
db=$(find /tmp/rp_t -name '*.db' | head -1)
[ -n "$db" ] && python /root/repo/tools/trace_extract.py $db /root/repo/gpurun_out/r04u_train64_side1.csv.gz | tail -1
