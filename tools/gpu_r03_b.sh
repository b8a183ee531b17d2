#!/bin/bash
# round 3, GPU session b: the tiled chain on long shards (whole-output sha256 against the reference run with the same
# plan), tile sizes, and the occupancy variants of k_ix_bucket / k_store.
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS} ) > gpurun_out/r03_b_$name.log 2>&1
  grep '^{' gpurun_out/r03_b_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['value'], c['ratio'], c['stage_ms'], 'sha_equal', c.get('parity_full_sha256_equal'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" || tail -5 gpurun_out/r03_b_$name.log
}
BENCH_ARGS="--shard-kb 1024" run tile128_1024k BROTLI_AMD_TILE_KB=128 BROTLI_AMD_TILE_LOG=1
grep "tile pass" gpurun_out/r03_b_tile128_1024k.log | tail -4
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline" run tile64_1024k BROTLI_AMD_TILE_KB=64
BENCH_ARGS="--shard-kb 2048 --no-cpu-baseline" run tile128_2048k BROTLI_AMD_TILE_KB=128
BENCH_ARGS="--shard-kb 512 --no-cpu-baseline" run tile128_512k BROTLI_AMD_TILE_KB=128
BENCH_ARGS="--shard-kb 256 --no-cpu-baseline" run tile64_256k BROTLI_AMD_TILE_KB=64
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline --workload silesia" run tile128_1024k_silesia BROTLI_AMD_TILE_KB=128 BROTLI_AMD_TILE_LOG=1
grep "tile pass" gpurun_out/r03_b_tile128_1024k_silesia.log | tail -6
BENCH_ARGS="--no-cpu-baseline" 
for v in lr6 lr5 lr8w5 sw2; do run $v BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so; done
