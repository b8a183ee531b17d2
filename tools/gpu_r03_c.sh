#!/bin/bash
# round 3, GPU session c: where the tiled chain's time goes and why shards leave it; k_ix_bucket LDS rows with parity.
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/r03_c_$name.log 2>&1
  grep '^{' gpurun_out/r03_c_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['value'], c['ratio'], c['stage_ms'])" || tail -5 gpurun_out/r03_c_$name.log
}
BENCH_ARGS="--shard-kb 1024" run tile128_1024k BROTLI_AMD_TILE_KB=128 BROTLI_AMD_TILE_LOG=1
grep -E "tile pass|tile stage|off the tiled" gpurun_out/r03_c_tile128_1024k.log | tail -14
BENCH_ARGS="--shard-kb 1024" run tile64_1024k BROTLI_AMD_TILE_KB=64 BROTLI_AMD_TILE_LOG=1
grep -E "tile pass|tile stage|off the tiled" gpurun_out/r03_c_tile64_1024k.log | tail -14
BENCH_ARGS="--shard-kb 256" run tile64_256k BROTLI_AMD_TILE_KB=64 BROTLI_AMD_TILE_LOG=1
grep -E "tile pass|tile stage|off the tiled" gpurun_out/r03_c_tile64_256k.log | tail -14
for v in lr5 lr4 lr6; do
  BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
  BENCH_ARGS="" run $v BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so
done
BENCH_ARGS="--shard-kb 1024" run lr5_1024k BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_lr5.so
