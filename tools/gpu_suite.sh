#!/bin/bash
# the whole GPU suite + smoke, log under gpurun_out/<tag>_pytest.log
TAG=${1:-suite}
OUT=/root/repo/gpurun_out; mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/${TAG}_pytest.log; tail -25 $OUT/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
