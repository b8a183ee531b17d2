#!/bin/bash
# PMC passes over the indexed parse (k_ix_* and k_chain), separate runs per counter set.
OUT=gpurun_out/ixpmc_${TAG:-x}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $set --kernel-trace -d /root/repo/$OUT/p$i -o p$i -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} ) > $OUT/p$i.log 2>&1
done
python tools/pmc_summary.py $OUT | grep -E "^DB|k_ix|k_chain|k_build|k_store" > $OUT/summary.txt
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
