#!/bin/bash
# One GPU-box session: gpu tests, smoke, bench, rocprofv3 kernel stats of the bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -x -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
( time timeout 900 python bench.py ${BENCH_ARGS} ) > gpurun_out/bench.log 2>&1
tail -4 gpurun_out/bench.log
rm -rf gpurun_out/prof_stats
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_stats -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/prof_stats.log 2>&1
tail -2 gpurun_out/prof_stats.log
python tools/pmc_summary.py gpurun_out/prof_stats > gpurun_out/prof_stats_summary.txt 2>&1
grep -E "KERNEL k_" gpurun_out/prof_stats_summary.txt
find gpurun_out/prof_stats -name "*.db" -size +20M -delete
