#!/bin/bash
# PMC passes over the parse kernel (one config), separate runs per counter set.
export PROBE_SKIP_PARITY=1 PROBE_MB=${PROBE_MB:-256} PROBE_SHARDS=${PROBE_SHARDS:-65536} PROBE_REPS=1
OUT=gpurun_out/pmc
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p$i -- python tools/gpu_parse_probe.py > $OUT/p$i.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- python tools/gpu_parse_probe.py > $OUT/stats.log 2>&1
find $OUT -name "*.csv" | head -50
python tools/pmc_summary.py $OUT
