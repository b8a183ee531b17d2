#!/bin/bash
# round 3, GPU session i: sweeps with the quick event check; tile tests
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS} ) > gpurun_out/r03_i_$name.log 2>&1
  grep '^{' gpurun_out/r03_i_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['value'], c['ratio'], c['stage_ms'], 'sha', c.get('parity_full_sha256_equal'))" || tail -5 gpurun_out/r03_i_$name.log
}
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline" run t1024_log BROTLI_AMD_TILE_LOG=1
grep -E "tile pass|tile stage|off the tiled" gpurun_out/r03_i_t1024_log.log | tail -9
BENCH_ARGS="--shard-kb 1024" run t1024
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline" run t1024_sg1 BROTLI_AMD_SWEEP_GROUPS=1
BENCH_ARGS="--shard-kb 1024 --no-cpu-baseline" run t1024_sg4 BROTLI_AMD_SWEEP_GROUPS=4
BENCH_ARGS="--shard-kb 512 --no-cpu-baseline" run t512
BENCH_ARGS="--shard-kb 2048 --no-cpu-baseline" run t2048
BENCH_ARGS="--shard-kb 4095 --no-cpu-baseline" run t4095
timeout 900 python -m pytest tests/test_gpu_zx_tiles.py -x -q -m gpu 2>&1 | tail -2
