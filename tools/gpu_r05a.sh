#!/bin/bash
# round 5, session a: GPU suite on the merged tree (f16x1 tests included), the default bench line with "also", encoder error table, config-5 step trace, TCC passes
TAG=${1:-r05a}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -4 $OUT/${TAG}_pytest.log
timeout 600 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; echo "bench rc=$?"; python tools/bench_summary.py $OUT/${TAG}_bench_default.json 30; tail -3 $OUT/${TAG}_bench_default.err
timeout 300 python tools/encoder_error_table.py $OUT/${TAG}_encoder_error_table.txt > /dev/null 2>$OUT/${TAG}_enc_table.err; tail -12 $OUT/${TAG}_encoder_error_table.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_t64
timeout 250 rocprofv3 --kernel-trace -d /tmp/rp_t64 -o p -- python /root/repo/bench.py --workload train64 --steps 3 --warmup 3 --no-roofline --cpu-faces 0 --traffic off > /tmp/rp_t64.log 2>&1
db=$(find /tmp/rp_t64 -name "*.db" | head -1)
[ -n "$db" ] && python /root/repo/tools/trace_extract.py $db $OUT/${TAG}_train64.csv.gz | tail -1 && python /root/repo/tools/step_timeline.py $OUT/${TAG}_train64.csv.gz > $OUT/${TAG}_timeline_train64.txt; head -30 $OUT/${TAG}_timeline_train64.txt | cut -c1-180
cd /root/repo
timeout 400 python tools/pmc_tcc.py full $OUT/${TAG}_pmc_tcc_full.txt > /dev/null 2>&1; head -40 $OUT/${TAG}_pmc_tcc_full.txt | cut -c1-200
