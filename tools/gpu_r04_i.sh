#!/bin/bash
# Round 4, session i (after the final run; measurements for the next round, nothing of it ships): SQ counters of every
# kernel of the bench's command, and the 1 MiB plan with two planner knobs (buckets per wave of k_ix_bucket, one-wave
# build / store).
ulimit -c 0
O=gpurun_out/r04i
mkdir -p $O
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$O/pmc$i -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O 2>/dev/null | grep -E "^DB|KERNEL k_|PMC k_chain|PMC k_build|PMC k_store|PMC k_ix_scatter|PMC k_ix_bucket" > $O/pmc_summary.txt
find $O -name "*.db" -delete
cat $O/pmc_summary.txt | tee $O/summary.txt
echo "== 1 MiB plan, planner knobs" | tee -a $O/summary.txt
for v in "default:" "bpw2:BROTLI_AMD_IX_BPW=2" "narrow:BROTLI_AMD_WIDE=0" "both:BROTLI_AMD_IX_BPW=2 BROTLI_AMD_WIDE=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 300 python bench.py --shard-kb 1024 --steps 3 --no-cpu-baseline > $O/bench_1024k_$name.json 2> $O/bench_1024k_$name.err
  python - "$name" <<'PY' | tee -a $O/summary.txt
import json, sys
name = sys.argv[1]
try:
    d = json.loads([ln for ln in open("gpurun_out/r04i/bench_1024k_%s.json" % name).read().splitlines() if ln.startswith("{")][-1])
    print("1 MiB plan", name, d["value"], d["ms_per_step"], d["config"]["stage_ms"])
except Exception as e:
    print("1 MiB plan", name, "unreadable", e)
PY
done
