#!/bin/bash
# Kernel traces (rocpd sqlite, analysed offline by tools/step_timeline.py) of the 128-frame shard step, infer256 and train64.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
run() {  # tag, bench args
  tag=$1; shift
  timeout 250 rocprofv3 --kernel-trace -d /tmp/rp_$tag -o p -- python /root/repo/bench.py "$@" --no-roofline --cpu-faces 0 --traffic off > /tmp/rp_$tag.log 2>&1
  db=$(find /tmp/rp_$tag -name '*.db' | head -1)
  if [ -n "$db" ]; then python /root/repo/tools/trace_extract.py $db $OUT/r04o_$tag.csv.gz; else echo "no db $tag"; tail -5 /tmp/rp_$tag.log; fi
  grep '^{' /tmp/rp_$tag.log | python /root/repo/tools/bench_summary.py /dev/stdin 0 | head -1
}
run shard128 --workload full --global-batch 128 --force-collective --steps 4 --warmup 3
run infer256 --workload infer256 --steps 4 --warmup 3
run train64 --workload train64 --steps 2 --warmup 2
