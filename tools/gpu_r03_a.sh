#!/bin/bash
# round 3, GPU session a: parity of the k_store / k_decode changes, baseline bench with the CPU reference beside it,
# occupancy variants of k_ix_bucket (LDS rows, bucket size) and of k_store (register budget).
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_decode.py tests/test_gpu_python_mirror.py -x -q -m gpu ) > gpurun_out/r03_a_pytest.log 2>&1
tail -3 gpurun_out/r03_a_pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r03_a_bench.log 2>&1
tail -1 gpurun_out/r03_a_bench.log | cut -c1-1200
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/r03_a_$name.log 2>&1
  tail -1 gpurun_out/r03_a_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['stage_ms'])" || tail -3 gpurun_out/r03_a_$name.log
}
for v in lr6 lr5 lr8w5 sw2; do run $v BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so; done
run t160 BROTLI_AMD_IX_TARGET=160
for v in lr4 lr4w6; do run ${v}_t160 BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so BROTLI_AMD_IX_TARGET=160; done
BENCH_ARGS="--shard-kb 512" run base_512k
BENCH_ARGS="--shard-kb 1024" run base_1024k
