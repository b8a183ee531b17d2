#!/bin/bash
# round 3, GPU session h: batches of a plain indexed job (index of batch b + 1 beside the chain of batch b)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 900 python bench.py --steps 5 --warmup 1 ${BENCH_ARGS} ) > gpurun_out/r03_h_$name.log 2>&1
  grep '^{' gpurun_out/r03_h_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['value'], d['ms_per_step'], c['stage_ms'], 'sha', c.get('parity_full_sha256_equal'))" || tail -5 gpurun_out/r03_h_$name.log
}
BENCH_ARGS="" run b4_full
for b in 1 2 3 6 8; do BENCH_ARGS="--no-cpu-baseline" run b$b BROTLI_AMD_BATCHES=$b; done
BENCH_ARGS="--no-cpu-baseline --workload silesia" run b4_silesia
BENCH_ARGS="--no-cpu-baseline --workload silesia" run b1_silesia BROTLI_AMD_BATCHES=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -m gpu 2>&1 | tail -2
