#!/bin/bash
# One GPU-box session: gpu tests, smoke, bench, rocprofv3 kernel stats of the bench, PMC traffic of k_parse4.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -x -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
( time timeout 900 python bench.py ${BENCH_ARGS} ) > gpurun_out/bench.log 2>&1
tail -4 gpurun_out/bench.log | cut -c1-1500
rm -rf gpurun_out/prof_stats
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_stats -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/prof_stats.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_stats > gpurun_out/prof_stats_summary.txt 2>&1
grep -E "KERNEL k_" gpurun_out/prof_stats_summary.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d_ -f1 | cut -c1-5)$(echo $set | wc -w)
  ( cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace -d /root/repo/gpurun_out/prof_pmc_$n -o bench -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/prof_pmc_$n.log 2>&1
  python tools/pmc_summary.py gpurun_out/prof_pmc_$n | grep -E "PMC k_parse4|KERNEL k_parse4" >> gpurun_out/prof_pmc_summary.txt
done
cat gpurun_out/prof_pmc_summary.txt
# quality 1 (BASELINE config 3: random bytes) and quality 9 (config 5): bench lines + kernel stats of the q1 job
( timeout 600 python bench.py --quality 1 --data random --steps 5 --warmup 1 ) > gpurun_out/bench_q1_random.log 2>&1; tail -1 gpurun_out/bench_q1_random.log | cut -c1-400
( timeout 600 python bench.py --quality 1 --data text --lgwin 18 --steps 3 --warmup 1 ) > gpurun_out/bench_q1_text_lgwin18.log 2>&1; tail -1 gpurun_out/bench_q1_text_lgwin18.log | cut -c1-400
( timeout 600 python bench.py --quality 1 --data text --feed-kb 512 --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_q1_text_cli.log 2>&1; tail -1 gpurun_out/bench_q1_text_cli.log | cut -c1-400
( timeout 600 python bench.py --quality 9 --lgwin 24 --shard-kb 512 --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_q9.log 2>&1; tail -1 gpurun_out/bench_q9.log | cut -c1-400
rm -rf gpurun_out/prof_stats_q1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_stats_q1 -o bench -- python /root/repo/bench.py --quality 1 --data random --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/prof_stats_q1.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_stats_q1 > gpurun_out/prof_stats_q1_summary.txt 2>&1
grep -E "KERNEL k_" gpurun_out/prof_stats_q1_summary.txt
find gpurun_out -name "*.db" -delete
BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --size-mb 256 > gpurun_out/bench_dist1.log 2>&1; tail -2 gpurun_out/bench_dist1.log | cut -c1-600
