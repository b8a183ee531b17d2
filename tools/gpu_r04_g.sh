#!/bin/bash
# Round 4, session g: the round's last word on the MI355X — the whole GPU suite, smoke(), the default bench line (both
# plans, the reference beside them, the stock calls), rocprofv3 kernel statistics of the bench's command, FETCH_SIZE /
# WRITE_SIZE of its kernels (separate passes), the other single-GPU BASELINE configurations with the reference beside each.
ulimit -c 0
O=gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $O/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $O/pytest.log)" | tee -a $O/summary.txt
echo "== smoke" | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?: $(tail -1 $O/smoke.log)" | tee -a $O/summary.txt
echo "== bench (default)" | tee -a $O/summary.txt
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
echo "== kernel statistics of the bench's command" | tee -a $O/summary.txt
rm -rf $O/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o bench -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/prof.log 2>&1
python tools/pmc_summary.py $O/prof > $O/kernel_stats_default_128KiB.txt 2>&1
grep -E "KERNEL k_" $O/kernel_stats_default_128KiB.txt | head -20 | tee -a $O/summary.txt
echo "== HBM counters of the bench's command (separate passes)" | tee -a $O/summary.txt
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$set
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$O/pmc_$set -o bench -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > $O/pmc_$set.log 2>&1
  python tools/pmc_summary.py $O/pmc_$set | grep -E "PMC k_ix|PMC k_chain|PMC k_build|PMC k_store" >> $O/pmc_summary.txt
done
cat $O/pmc_summary.txt | tee -a $O/summary.txt
echo "== the other single-GPU configurations" | tee -a $O/summary.txt
( time timeout 400 python bench.py --quality 1 --data random ) > $O/bench_q1_random.json 2> $O/bench_q1_random.err
echo "q1 random rc $?" | tee -a $O/summary.txt
( time timeout 600 python bench.py --workload silesia --steps 3 ) > $O/bench_mix.json 2> $O/bench_mix.err
echo "mix rc $?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --quality 9 --lgwin 24 --shard-kb 384 --steps 3 ) > $O/bench_q9.json 2> $O/bench_q9.err
echo "q9 rc $?" | tee -a $O/summary.txt
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmcmix_$set
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$O/pmcmix_$set -o bench -- python /root/repo/bench.py --workload silesia --steps 1 --warmup 0 --no-cpu-baseline ) > $O/pmcmix_$set.log 2>&1
  python tools/pmc_summary.py $O/pmcmix_$set | grep -E "PMC k_ix_bucket|PMC k_chain" >> $O/pmc_mix_summary.txt
done
cat $O/pmc_mix_summary.txt | tee -a $O/summary.txt
find $O -name "*.db" -delete
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g/bench*.json")):
    try:
        d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
        c = d["config"]; b = d.get("cpu_baseline") or {}
        print(f.split("/")[-1], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms |", c.get("stage_ms"), "| ratio", c.get("ratio"),
              "| cpu", b.get("value"), b.get("cores"), "| sha", c.get("parity_full_sha256_equal", c.get("spot_check_first_16MiB_bit_exact")),
              "| roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
        for p in c.get("plans", []):
            print("   plan", {k: p.get(k) for k in ("shard_KiB", "MBps", "ratio", "reference_same_plan_MBps", "x_reference_same_plan", "sha256_equal_reference", "error")})
        sc = c.get("stock_call_no_plan")
        if sc: print("   stock", {k: sc.get(k) for k in ("MBps", "reference_1core_MBps", "bytes_equal_reference")}, sc.get("whole_input"))
        if c.get("end_to_end_abi"): print("   abi", c["end_to_end_abi"])
    except Exception as e:
        print(f, "unreadable", e)
PY
