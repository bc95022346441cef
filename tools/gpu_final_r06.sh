#!/bin/bash
# Round-6 evidence for the SHIPPED commit (run last): GPU suite + smoke, rocprofv3 --kernel-trace --stats of config 4 in the timed-region schedule AND in the serial
# schedule (--no-overlap, one whole-batch chain: the state roofline.achieved / frac describe) and of config 5, the per-kernel averages bench.py quotes as
# roofline.rocprofv3 / roofline.rocprofv3_serial, then the driver's own command (`python bench.py`: headline with roofline + PMC traffic + cpu_baseline, every other
# BASELINE config under "also", each with its own cpu_baseline), the TCC passes of the dominant kernel, the per-rank shard lines.
TAG=${1:-r06z}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log | cut -c1-200
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee $OUT/${TAG}_smoke.txt
cd /tmp && export TMPDIR=/tmp
prof() {   # name, extra env, bench args
  local name=$1 envs=$2; shift 2
  rm -rf /tmp/rp_$name
  env $envs timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o p -- python /root/repo/bench.py "$@" --steps 2 --warmup 1 --no-roofline --cpu-faces 0 --no-also > /tmp/rp_$name.log 2>&1
  db=$(find /tmp/rp_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python /root/repo/tools/rocprof_summary.py $db $OUT/${TAG}_kernel_stats_$name.txt "$envs rocprofv3 --kernel-trace --stats -- python bench.py $* --steps 2 --warmup 1 --no-roofline --cpu-faces 0 --no-also" $OUT/rocprofv3_kernel_avg_$name.json | head -10 | cut -c1-130; else echo "no db $name"; tail -3 /tmp/rp_$name.log; fi
}
prof full "SMIRK_X=0" --workload full
prof full_serial "SMIRK_GEN_SPLIT_CHAINS=0" --workload full --no-overlap
prof train64 "SMIRK_X=0" --workload train64
cd /root/repo
# the per-launch averages must be in profiles/ BEFORE the bench line is produced (roofline.rocprofv3* read them; same kernel sources by construction)
for n in full full_serial train64; do cp $OUT/rocprofv3_kernel_avg_$n.json profiles/rocprofv3_kernel_avg_$n.json 2>/dev/null; done
( time timeout 1100 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err ) 2>&1 | grep real; python tools/bench_summary.py $OUT/${TAG}_bench_default.json 30
cp gpurun_out/pmc_traffic_full.json $OUT/${TAG}_pmc_traffic_full.json 2>/dev/null
timeout 400 python tools/pmc_tcc.py full $OUT/${TAG}_pmc_tcc_full.txt > /dev/null 2>&1; head -24 $OUT/${TAG}_pmc_tcc_full.txt | cut -c1-160
for g in 128 256 512 1024; do
  timeout 200 python bench.py --global-batch $g --force-collective --steps 20 --warmup 5 --traffic off --cpu-faces 0 --no-roofline --no-also > $OUT/${TAG}_shard_$g.json 2> /dev/null; python tools/bench_summary.py $OUT/${TAG}_shard_$g.json 1
done
