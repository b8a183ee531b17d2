#!/bin/bash
# Round 4, session e: the levelled window search of k_ix_bucket (pairs spread over the lanes) against the slot loop
# (F0 = the commit before, F1 = every pair through the rounds, F2 = the product: nearest candidate local first), its SQ
# counters, the parity file on the new kernel, bench lines (text, the mix with the spree steps of the chain).
ulimit -c 0
O=gpurun_out/r04e
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest (parity file)" | tee $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > $O/pytest_parity.log 2>&1
echo "pytest parity rc $?: $(tail -1 $O/pytest_parity.log)" | tee -a $O/summary.txt
echo "== index variants (k_ix_* alone, 1 GiB text)" | tee -a $O/summary.txt
TAG=F0 BROTLI_AMD_HIP_LIB=$PWD/build/var/ixF0.so PROBE_SHARDS=131072,1048576 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
TAG=F1 BROTLI_AMD_HIP_LIB=$PWD/build/var/ixF1.so PROBE_SHARDS=131072,1048576 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
TAG=F2 PROBE_SHARDS=131072,1048576 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
echo "== counters of the index kernels (product library, 128 KiB shards)" | tee -a $O/summary.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  ( cd /tmp && TAG=pmc PROBE_SHARDS=131072 timeout 300 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$O/pmc$i -o p -- python /root/repo/tools/gpu_ix_only.py ) > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O 2>/dev/null | grep -E "^DB|k_ix_bucket" > $O/pmc_summary.txt
find $O -name "*.db" -delete
cat $O/pmc_summary.txt | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
timeout 300 python bench.py --shard-kb 1024 --steps 3 --no-cpu-baseline > $O/bench_1024k.json 2> $O/bench_1024k.err
echo "bench 1 MiB rc $?" | tee -a $O/summary.txt
timeout 400 python bench.py --workload silesia --steps 3 --no-cpu-baseline > $O/bench_mix.json 2> $O/bench_mix.err
echo "bench mix rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04e/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("stage_ms"), d["config"].get("device_round_trip", {}).get("equal_to_input"))
    except Exception as e:
        print(f, "unreadable", e)
PY
