#!/bin/bash
# round 3, GPU session e: sweeps after the rework (one tile per wave, replay through the open gate), big buckets without
# gathers, tile tests.
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS} ) > gpurun_out/r03_e_$name.log 2>&1
  grep '^{' gpurun_out/r03_e_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['value'], c['ratio'], c['stage_ms'], 'sha_equal', c.get('parity_full_sha256_equal'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" || tail -5 gpurun_out/r03_e_$name.log
}
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline" run tile128_1024k_log BROTLI_AMD_TILE_KB=128 BROTLI_AMD_TILE_LOG=1
grep -E "tile pass|tile stage|off the tiled" gpurun_out/r03_e_tile128_1024k_log.log | tail -9
BENCH_ARGS="--shard-kb 1024" run tile128_1024k BROTLI_AMD_TILE_KB=128
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline" run tile64_1024k BROTLI_AMD_TILE_KB=64
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline" run tile128_1024k_sg2 BROTLI_AMD_TILE_KB=128 BROTLI_AMD_SWEEP_GROUPS=2
BENCH_ARGS="--shard-kb 512 --no-cpu-baseline" run tile128_512k BROTLI_AMD_TILE_KB=128
BENCH_ARGS="--shard-kb 2048 --no-cpu-baseline" run tile128_2048k BROTLI_AMD_TILE_KB=128
BENCH_ARGS="--no-cpu-baseline" run default_128k
timeout 900 python -m pytest tests/test_gpu_zx_tiles.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
