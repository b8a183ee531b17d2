#!/bin/bash
# round 3, GPU session j: what the driver runs at round end — the whole `-m gpu` suite, smoke(), the default bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03_j_pytest_gpu.log 2>&1
tail -4 gpurun_out/r03_j_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_j_smoke.log 2>&1; tail -4 gpurun_out/r03_j_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r03_j_bench.log 2>&1
grep '^{' gpurun_out/r03_j_bench.log | tail -1 | cut -c1-700
