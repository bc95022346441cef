#!/bin/bash
# round 6: counter-traffic tables of the other BASELINE configs (what the `also` children quote through --traffic file), then the driver's command once more
TAG=${1:-r06z}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
for wl in infer256 flame512 train64; do
  timeout 600 python bench.py --workload $wl --steps 5 --warmup 2 --traffic measure --cpu-faces 0 --no-also > $OUT/${TAG}_bench_${wl}_pmc.json 2> $OUT/${TAG}_bench_${wl}_pmc.err; python tools/bench_summary.py $OUT/${TAG}_bench_${wl}_pmc.json 2
  cp gpurun_out/pmc_traffic_$wl.json profiles/pmc_traffic_$wl.json 2>/dev/null
done
( time timeout 1100 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err ) 2>&1 | grep real; python tools/bench_summary.py $OUT/${TAG}_bench_default.json 8
cp gpurun_out/pmc_traffic_full.json $OUT/${TAG}_pmc_traffic_full.json 2>/dev/null
