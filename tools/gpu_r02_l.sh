#!/bin/bash
# Round 2, GPU session L: PMC passes of the bench (issue / wait / LDS counters, then FETCH / WRITE separately).
OUT=gpurun_out/l_pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$OUT/p$i -o p$i -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > $OUT/p$i.log 2>&1
done
python tools/pmc_summary.py $OUT | grep -E "^DB|PMC" > $OUT/summary.txt
find $OUT -name "*.db" -delete
grep -E "k_ix_bucket|k_chain|k_ix_scatter" $OUT/summary.txt
