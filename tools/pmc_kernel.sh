#!/bin/bash
# usage: tools/pmc_kernel.sh <conv_sweep --only filter> <kernel-name substring> <out file>   (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/$3
: > $OUT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc -o p -- python /root/repo/tools/conv_sweep.py --only "$1" --iters 3 > /tmp/pmc.log 2>&1
  db=$(find /tmp/pmc -name "*.db" | head -1)
  for c in $grp; do
    echo "== $c" >> $OUT
    python /root/repo/tools/pmc_summary.py $db $c 2>&1 | grep -i "$2" | head -2 >> $OUT
  done
done
cat $OUT
