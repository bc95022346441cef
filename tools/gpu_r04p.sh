#!/bin/bash
# Hardware-queue count: the trace of the 128-frame step (profiles/r04o_*) shows the three backbone streams starting only when a packet of the generator's stream
# retires — 8+ HIP streams share the runtime's 4 hardware queues.  Same box: GPU_MAX_HW_QUEUES unset / 8 / 16 on every bench workload, then the trace with 8.
cd /root/repo
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for wl in "shard128 --workload full --global-batch 128 --force-collective --steps 30 --warmup 5" "shard256 --workload full --global-batch 256 --force-collective --steps 20 --warmup 5" \
            "full1024 --workload full --steps 12 --warmup 3" "infer256 --workload infer256 --steps 40 --warmup 10" "train64 --workload train64 --steps 10 --warmup 3"; do
    set -- $wl; tag=$1; shift
    python bench.py "$@" --cpu-faces 0 --traffic off --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r04p_${tag}_q$q.json
    echo "queues=$q $tag $(python tools/bench_summary.py gpurun_out/r04p_${tag}_q$q.json 0 | head -1)"
  done
done
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace -d /tmp/rp_q8 -o p -- python /root/repo/bench.py --workload full --global-batch 128 --force-collective --steps 4 --warmup 3 --no-roofline --cpu-faces 0 --traffic off > /tmp/rp_q8.log 2>&1
db=$(find /tmp/rp_q8 -name '*.db' | head -1)
[ -n "$db" ] && python /root/repo/tools/trace_extract.py $db /root/repo/gpurun_out/r04p_shard128_q8.csv.gz | tail -1
