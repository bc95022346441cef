#!/bin/bash
# Round-end measurement session on the GPU box (run through gpurun): tests, the four bench lines, rocprofv3 kernel stats, smoke.
TAG=${1:-r03z}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log
tail -3 $OUT/${TAG}_pytest.log
for wl in full infer256 flame512 train64; do
  extra=""; [ "$wl" = "train64" ] && extra="--steps 5 --warmup 2 --cpu-faces 4"
  [ "$wl" = "full" ] && extra="--steps 5 --warmup 2"
  timeout 1200 python bench.py --workload $wl $extra > $OUT/${TAG}_bench_${wl}.json 2> $OUT/${TAG}_bench_${wl}.err
  echo "bench $wl rc=$? $(python -c "import json;j=json.load(open('$OUT/${TAG}_bench_${wl}.json'));r=j['roofline'];print(round(j['value'],1),j['unit'],round(j['ms_per_step'],2),'ms/step |',r['kernel'],r['bound'],round(r['frac'],4),'traffic',r['traffic'],'| cpu',j['cpu_baseline'] and round(j['cpu_baseline']['value'],2))" 2>&1 | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
for wl in full train64; do
  rm -rf /tmp/rp_$wl
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_$wl -o p -- python /root/repo/bench.py --workload $wl --steps 2 --warmup 1 --no-roofline --cpu-faces 0 > /tmp/rp_$wl.log 2>&1
  db=$(find /tmp/rp_$wl -name "*.db" | head -1)
  if [ -n "$db" ]; then python /root/repo/tools/rocprof_summary.py $db $OUT/${TAG}_kernel_stats_${wl}.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 2 --warmup 1 --no-roofline --cpu-faces 0" | head -14 | cut -c1-120; else echo "no db for $wl"; ls -R /tmp/rp_$wl | head; tail -3 /tmp/rp_$wl.log; fi
done
cd /root/repo
cp gpurun_out/pmc_traffic_full.json $OUT/${TAG}_pmc_traffic_full.json 2>/dev/null
cp gpurun_out/pmc_traffic_train64.json $OUT/${TAG}_pmc_traffic_train64.json 2>/dev/null
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 600 python bench.py --flame-basis smooth --traffic off --cpu-faces 0 > $OUT/${TAG}_bench_full_smooth_basis.json 2>/dev/null; python -c "
import json; j=json.load(open('$OUT/${TAG}_bench_full_smooth_basis.json')); print('full, smooth FLAME basis', round(j['value'],1), round(j['ms_per_step'],2))"
