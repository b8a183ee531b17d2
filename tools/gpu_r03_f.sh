#!/bin/bash
# round 3, GPU session f: the evidence run.  One line per single-GPU BASELINE configuration with the reference beside it
# (bench.py --all-configs), the long-shard plan, rocprofv3 kernel stats of the default command and of the long-shard
# command, counter calibration + PMC passes for the traffic of k_ix_bucket.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python bench.py --all-configs ) > gpurun_out/r03_f_all_configs.log 2>&1
grep '^{' gpurun_out/r03_f_all_configs.log | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    if 'error' in d: print(d['config_label'], 'ERROR', d['error'][-300:]); continue
    c = d['config']; b = d.get('cpu_baseline') or {}
    print(d['config_label'], '|', d['value'], d['unit'], '| ratio', c.get('ratio'), '| cpu', b.get('value'), b.get('cores'), '| sha', c.get('parity_full_sha256_equal', c.get('spot_check_first_16MiB_bit_exact')), '| roofline', d['roofline']['kernel'], d['roofline']['frac'])
"
( time timeout 900 python bench.py --shard-kb 1024 ) > gpurun_out/r03_f_bench_1024k.log 2>&1
grep '^{' gpurun_out/r03_f_bench_1024k.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('1024k', d['value'], c['ratio'], c['stage_ms'], 'sha', c.get('parity_full_sha256_equal'), 'cpu', d['cpu_baseline']['value'], 'e2e', c.get('end_to_end_abi'))"
for tag in "default:" "1024k:--shard-kb 1024"; do
  name=${tag%%:*}; extra=${tag#*:}
  rm -rf gpurun_out/prof_$name
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$name -o bench -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra ) > gpurun_out/r03_f_prof_$name.log 2>&1
  python tools/pmc_summary.py gpurun_out/prof_$name > gpurun_out/r03_f_kernel_stats_$name.txt 2>&1
  grep -E "KERNEL k_" gpurun_out/r03_f_kernel_stats_$name.txt | head -30
done
# counters: calibration on known byte counts, then the bench
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_calib_$set gpurun_out/pmc_bench_$set
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d /root/repo/gpurun_out/pmc_calib_$set -o calib -- /root/repo/build/write_calib 1024 ) > gpurun_out/r03_f_pmc_calib_$set.log 2>&1
  ( cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace -d /root/repo/gpurun_out/pmc_bench_$set -o bench -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > gpurun_out/r03_f_pmc_bench_$set.log 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_calib_$set | grep -E "PMC calib" >> gpurun_out/r03_f_pmc_summary.txt
  python tools/pmc_summary.py gpurun_out/pmc_bench_$set | grep -E "PMC k_ix|PMC k_chain|PMC k_build|PMC k_store" >> gpurun_out/r03_f_pmc_summary.txt
done
cat gpurun_out/r03_f_pmc_summary.txt
find gpurun_out -name "*.db" -size +20M -delete
