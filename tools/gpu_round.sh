#!/bin/bash
# One GPU-box session: GPU test suite, the three bench workloads (with their rocprofv3 PMC traffic passes), kernel-trace stats.
#   usage (via gpurun): bash tools/gpu_round.sh <tag> [skip-tests]
TAG=${1:-r02a}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
if [ "$2" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
  tail -5 $OUT/${TAG}_pytest.log
fi
for wl in full infer256 flame512; do
  timeout 900 python bench.py --workload $wl --steps 5 --warmup 2 > $OUT/${TAG}_bench_${wl}.json 2> $OUT/${TAG}_bench_${wl}.err
  echo "bench $wl rc=$?"; head -c 600 $OUT/${TAG}_bench_${wl}.json; echo
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python /root/repo/bench.py --workload full --batch 128 --steps 3 --warmup 1 --cpu-faces 0 --traffic off > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python /root/repo/tools/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload full --batch 128 --steps 3 --warmup 1 --cpu-faces 0 --traffic off (5 passes of 128 frames incl. warm-up + instrumented)" > /dev/null
head -30 $OUT/${TAG}_kernel_stats.txt
