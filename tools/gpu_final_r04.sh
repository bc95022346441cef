#!/bin/bash
# Round-4 evidence for the SHIPPED commit (run last): GPU suite + smoke, the four bench lines (the default `python bench.py` run exactly as the driver issues it,
# with roofline + PMC traffic + cpu_baseline), the per-rank shard lines, rocprofv3 --kernel-trace --stats of config 4 and config 5.  CPU-baseline legs stay
# at the bounded default sample (no all-cores run on the GPU lease).
TAG=${1:-r04z}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -3 $OUT/${TAG}_pytest.log
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee $OUT/${TAG}_smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_full.json 2> $OUT/${TAG}_bench_full.err; echo "full rc=$?"; python tools/bench_summary.py $OUT/${TAG}_bench_full.json 30
cp gpurun_out/pmc_traffic_full.json $OUT/${TAG}_pmc_traffic_full.json 2>/dev/null
timeout 300 python bench.py --workload infer256 --steps 20 --warmup 5 > $OUT/${TAG}_bench_infer256.json 2> $OUT/${TAG}_bench_infer256.err; python tools/bench_summary.py $OUT/${TAG}_bench_infer256.json 8
timeout 300 python bench.py --workload flame512 --steps 50 --warmup 10 > $OUT/${TAG}_bench_flame512.json 2> $OUT/${TAG}_bench_flame512.err; python tools/bench_summary.py $OUT/${TAG}_bench_flame512.json 4
timeout 400 python bench.py --workload train64 --steps 10 --warmup 3 > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_bench_train64.err; python tools/bench_summary.py $OUT/${TAG}_bench_train64.json 10
bash tools/gpu_shards.sh ${TAG}
cd /tmp && export TMPDIR=/tmp
for wl in full train64; do
  rm -rf /tmp/rp_$wl
  timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/rp_$wl -o p -- python /root/repo/bench.py --workload $wl --steps 2 --warmup 1 --no-roofline --cpu-faces 0 > /tmp/rp_$wl.log 2>&1
  db=$(find /tmp/rp_$wl -name "*.db" | head -1)
  if [ -n "$db" ]; then python /root/repo/tools/rocprof_summary.py $db $OUT/${TAG}_kernel_stats_$wl.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2 --warmup 1 --no-roofline --cpu-faces 0" | head -14 | cut -c1-130; else echo "no db $wl"; tail -3 /tmp/rp_$wl.log; fi
done
cd /root/repo
# the clock pass runs the three backbones on ONE stream: GRBM_GUI_ACTIVE is a device-wide counter, so a kernel that overlaps kernels of other streams is charged
# their cycles too (round 3's 589 M cycles for a 0.55 ms mbconv_image launch, rows above 2.4 GHz)
SMIRK_ENCODER_SERIAL=1 timeout 200 python tools/pmc_clock.py full $OUT/${TAG}_pmc_clock_full.txt > /dev/null 2>&1; head -24 $OUT/${TAG}_pmc_clock_full.txt 2>/dev/null | cut -c1-150
# kernel trace of the 128-frame shard step with the shipped defaults (one hardware queue per stream): which stream sits on which queue, when each stream starts
cd /tmp
rm -rf /tmp/rp_shard
timeout 250 rocprofv3 --kernel-trace -d /tmp/rp_shard -o p -- python /root/repo/bench.py --workload full --global-batch 128 --force-collective --steps 4 --warmup 3 --no-roofline --cpu-faces 0 --traffic off > /tmp/rp_shard.log 2>&1
db=$(find /tmp/rp_shard -name "*.db" | head -1)
[ -n "$db" ] && python /root/repo/tools/trace_extract.py $db $OUT/${TAG}_shard128.csv.gz | tail -1 && python /root/repo/tools/step_timeline.py $OUT/${TAG}_shard128.csv.gz > $OUT/${TAG}_timeline_shard128.txt && head -22 $OUT/${TAG}_timeline_shard128.txt | cut -c1-200
