#!/bin/bash
# round 3, GPU session o: the suite and the long-shard bench line with 64 KiB tiles as the default
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 800 python -m pytest tests -x -q -m gpu ) > gpurun_out/r03_o_pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/r03_o_pytest_gpu.log | tail -2
( timeout 300 python bench.py --shard-kb 1024 --steps 3 ) > gpurun_out/r03_o_bench_1024k.log 2>&1
grep '^{' gpurun_out/r03_o_bench_1024k.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('1024k', d['value'], c['ratio'], c['stage_ms'], 'sha', c.get('parity_full_sha256_equal'), 'cpu', d['cpu_baseline']['value'])"
