#!/bin/bash
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/ixv_$name.log 2>&1
  tail -1 gpurun_out/ixv_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['stage_ms'])" || tail -3 gpurun_out/ixv_$name.log
}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
run wide8
for v in ww5 ww6; do run $v BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so; done
BENCH_ARGS="--shard-kb 256" run wide8_256k
BENCH_ARGS="--shard-kb 512" run wide8_512k
