#!/bin/bash
# config 5: where do the ~1500 small torch-side launches of a training step come from?  ATen census with Python call sites + a named kernel trace.
cd /root/repo
timeout 300 python tools/train_ops_census.py > gpurun_out/r04s_train_ops_census.txt 2> gpurun_out/r04s_census.err; tail -3 gpurun_out/r04s_census.err; head -50 gpurun_out/r04s_train_ops_census.txt
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace -d /tmp/rp_t -o p -- python /root/repo/bench.py --workload train64 --steps 2 --warmup 2 --no-roofline --cpu-faces 0 --traffic off > /tmp/rp_t.log 2>&1
db=$(find /tmp/rp_t -name '*.db' | head -1)
[ -n "$db" ] && python /root/repo/tools/trace_extract.py $db /root/repo/gpurun_out/r04s_train64.csv.gz | tail -1
