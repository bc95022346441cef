#!/bin/bash
# Round-5 evidence for the SHIPPED commit (run last): GPU suite + smoke, the driver's own command (`python bench.py`: headline with roofline + PMC traffic +
# cpu_baseline, the other BASELINE configs as stand-alone children under "also"), rocprofv3 --kernel-trace --stats of config 4 and config 5 (+ the per-kernel
# averages bench.py quotes as roofline.rocprofv3), the TCC passes of the dominant kernel.
TAG=${1:-r05z}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -3 $OUT/${TAG}_pytest.log
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee $OUT/${TAG}_smoke.txt
cd /tmp && export TMPDIR=/tmp
for wl in full train64; do
  rm -rf /tmp/rp_$wl
  timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/rp_$wl -o p -- python /root/repo/bench.py --workload $wl --steps 2 --warmup 1 --no-roofline --cpu-faces 0 --no-also > /tmp/rp_$wl.log 2>&1
  db=$(find /tmp/rp_$wl -name "*.db" | head -1)
  if [ -n "$db" ]; then python /root/repo/tools/rocprof_summary.py $db $OUT/${TAG}_kernel_stats_$wl.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2 --warmup 1 --no-roofline --cpu-faces 0 --no-also" $OUT/rocprofv3_kernel_avg_$wl.json | head -12 | cut -c1-130; else echo "no db $wl"; tail -3 /tmp/rp_$wl.log; fi
done
cd /root/repo
# the per-launch averages must be in profiles/ BEFORE the bench line is produced (roofline.rocprofv3 reads them; same kernel sources by construction)
cp $OUT/rocprofv3_kernel_avg_full.json profiles/rocprofv3_kernel_avg_full.json 2>/dev/null; cp $OUT/rocprofv3_kernel_avg_train64.json profiles/rocprofv3_kernel_avg_train64.json 2>/dev/null
( time timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err ) 2>&1 | grep real; python tools/bench_summary.py $OUT/${TAG}_bench_default.json 30
cp gpurun_out/pmc_traffic_full.json $OUT/${TAG}_pmc_traffic_full.json 2>/dev/null
timeout 400 python tools/pmc_tcc.py full $OUT/${TAG}_pmc_tcc_full.txt > /dev/null 2>&1; head -24 $OUT/${TAG}_pmc_tcc_full.txt | cut -c1-160
