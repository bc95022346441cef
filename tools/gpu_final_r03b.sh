#!/bin/bash
# Short follow-up to gpu_final_r03.sh (whose all-cores CPU baselines ate the session): kernel stats of the shipped library, the train64 line, smoke.
TAG=${1:-r03y}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_full
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/rp_full -o p -- python /root/repo/bench.py --workload full --steps 2 --warmup 1 --no-roofline --cpu-faces 0 > /tmp/rp_full.log 2>&1
db=$(find /tmp/rp_full -name "*.db" | head -1)
if [ -n "$db" ]; then python /root/repo/tools/rocprof_summary.py $db $OUT/${TAG}_kernel_stats_full.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload full --steps 2 --warmup 1 --no-roofline --cpu-faces 0" | head -12 | cut -c1-120; else echo "no db"; tail -3 /tmp/rp_full.log; fi
cd /root/repo
timeout 150 python bench.py --workload train64 --steps 5 --warmup 2 --cpu-faces 0 --traffic file > $OUT/${TAG}_bench_train64.json 2> $OUT/${TAG}_bench_train64.err
echo "bench train64 rc=$? $(python -c "import json;j=json.load(open('$OUT/${TAG}_bench_train64.json'));r=j['roofline'];print(round(j['value'],1),j['unit'],round(j['ms_per_step'],2),'ms/step |',r['kernel'],r['bound'],round(r['frac'],4),'traffic',r['traffic'])" 2>&1 | tail -1)"
timeout 100 python __graft_entry__.py --smoke 2>&1 | tail -1
cd /tmp
rm -rf /tmp/rp_train64
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/rp_train64 -o p -- python /root/repo/bench.py --workload train64 --steps 2 --warmup 1 --no-roofline --cpu-faces 0 > /tmp/rp_train64.log 2>&1
db=$(find /tmp/rp_train64 -name "*.db" | head -1)
if [ -n "$db" ]; then python /root/repo/tools/rocprof_summary.py $db $OUT/${TAG}_kernel_stats_train64.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload train64 --steps 2 --warmup 1 --no-roofline --cpu-faces 0" | head -8 | cut -c1-120; else echo "no db train64"; tail -3 /tmp/rp_train64.log; fi
