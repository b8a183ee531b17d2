#!/bin/bash
# Round 2, GPU session F: the whole -m gpu suite, smoke, the default bench line (with the CPU baseline and the ABI end-to-end number), kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -x -q -m gpu ) > gpurun_out/f_pytest_gpu.log 2>&1
tail -4 gpurun_out/f_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/f_smoke.log 2>&1; tail -2 gpurun_out/f_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/f_bench.log 2>&1
tail -1 gpurun_out/f_bench.log | cut -c1-3500
rm -rf gpurun_out/f_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/f_prof -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/f_prof.log 2>&1
python tools/pmc_summary.py gpurun_out/f_prof > gpurun_out/f_prof_summary.txt 2>&1
grep -E "KERNEL k_|KERNEL void" gpurun_out/f_prof_summary.txt
find gpurun_out -name "*.db" -delete
