#!/bin/bash
# Round 4, session d: k_ix_bucket with its loads in flight together (5 and 4 waves per SIMD, 2 / 4 / 8 buckets per wave),
# its SQ counters and HBM traffic, the giant-bucket lists and the refined taint rule on the Silesia-style mix, quality 9
# alone in a process, the bench line with both plans.
ulimit -c 0
O=gpurun_out/r04d
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest (parity files)" | tee $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > $O/pytest_parity.log 2>&1
echo "pytest parity rc $?: $(tail -1 $O/pytest_parity.log)" | tee -a $O/summary.txt
echo "== index variants (k_ix_* alone, 1 GiB)" | tee -a $O/summary.txt
TAG=P5 PROBE_SHARDS=131072,1048576 PROBE_BPW=0,2,4,8 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
TAG=P4 BROTLI_AMD_HIP_LIB=$PWD/build/var/ixw4.so PROBE_SHARDS=131072,1048576 PROBE_BPW=0,4,8 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
TAG=P5 PROBE_KIND=mix PROBE_SHARDS=131072 PROBE_BPW=0 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
TAG=P5nogiant BROTLI_AMD_NO_GIANT_LISTS=1 PROBE_KIND=mix PROBE_SHARDS=131072 PROBE_BPW=0 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
echo "== counters of the index kernels (product library, 128 KiB shards)" | tee -a $O/summary.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && TAG=pmc PROBE_SHARDS=131072 timeout 300 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$O/pmc$i -o p -- python /root/repo/tools/gpu_ix_only.py ) > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O 2>/dev/null | grep -E "^DB|k_ix" > $O/pmc_summary.txt
find $O -name "*.db" -delete
cat $O/pmc_summary.txt | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
timeout 400 python bench.py --workload silesia --steps 3 > $O/bench_mix.json 2> $O/bench_mix.err
echo "bench mix rc $?" | tee -a $O/summary.txt
timeout 300 python bench.py --quality 9 --lgwin 24 --shard-kb 512 --steps 2 --no-cpu-baseline > $O/bench_q9.json 2> $O/bench_q9.err
echo "bench q9 rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04d/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("stage_ms"), d["config"].get("parity_full_sha256_equal"))
        for p in d["config"].get("plans", []):
            print("   plan", {k: p.get(k) for k in ("shard_KiB", "MBps", "ratio", "reference_same_plan_MBps", "x_reference_same_plan", "sha256_equal_reference", "error")})
        sc = d["config"].get("stock_call_no_plan")
        if sc: print("   stock", {k: sc.get(k) for k in ("MBps", "reference_1core_MBps", "bytes_equal_reference")}, sc.get("whole_input"))
    except Exception as e:
        print(f, "unreadable", e)
PY
