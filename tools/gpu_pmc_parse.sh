#!/bin/bash
# PMC passes over the parse kernel (one config), separate runs per counter set.
export PROBE_SKIP_PARITY=1 PROBE_MB=${PROBE_MB:-1024} PROBE_SHARDS=${PROBE_SHARDS:-65536} PROBE_REPS=1
OUT=gpurun_out/pmc_${TAG:-x}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p$i -- python tools/gpu_parse_probe.py > $OUT/p$i.log 2>&1
done
python tools/pmc_summary.py $OUT | grep -E "^DB|k_parse" > $OUT/summary.txt
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
