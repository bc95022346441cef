#!/bin/bash
# SQ counters of one weight-gradient layer: tools/pmc_wgrad.sh <H,Cout,Cin,k> <kernel-name substring> <out file> [SMIRK_WGRAD_F16 mode]   (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/$3
export SMIRK_WGRAD_F16=${4:-2}
echo "# layer $1, SMIRK_WGRAD_F16=$SMIRK_WGRAD_F16, B=64; per-dispatch means over the sweep's 7 launches (rocprofv3 --pmc <group> --kernel-trace, one run per group)" > $OUT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc -o p -- python /root/repo/tools/wgrad_sweep.py 64 $1 > /tmp/pmc.log 2>&1
  db=$(find /tmp/pmc -name "*.db" | head -1)
  if [ -z "$db" ]; then echo "!! no database for group: $grp" >> $OUT; tail -5 /tmp/pmc.log >> $OUT; continue; fi
  for c in $grp; do
    echo "== $c" >> $OUT
    python /root/repo/tools/pmc_summary.py $db $c 2>&1 | grep -i "$2" | head -2 >> $OUT
  done
done
cat $OUT
