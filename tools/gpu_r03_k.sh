#!/bin/bash
# round 3, GPU session k: a stock caller's one-shot quality-5 call (tests + timing), the fuzz slice on the GPU
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_abi.py -x -q -m gpu -k "one_shot" 2>&1 | tail -2
( timeout 900 python bench.py --shard-kb 1024 --steps 3 ) > gpurun_out/r03_k_bench_1024k.log 2>&1
grep '^{' gpurun_out/r03_k_bench_1024k.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('1024k', d['value'], c['ratio'], 'sha', c.get('parity_full_sha256_equal'), 'stock', c.get('stock_call_no_plan'))"
( timeout 900 python bench.py --lgwin 24 --shard-kb 1024 --steps 2 --size-mb 256 ) > gpurun_out/r03_k_bench_lgwin24.log 2>&1
grep '^{' gpurun_out/r03_k_bench_lgwin24.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('lgwin24', d['value'], c['ratio'], 'sha', c.get('parity_full_sha256_equal'), 'stock', c.get('stock_call_no_plan'))"
