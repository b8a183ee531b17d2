#!/bin/bash
# round 3, GPU session n: tile size and warm-up length on 1 MiB shards
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --shard-kb 1024 ) > gpurun_out/r03_n_$name.log 2>&1
  grep '^{' gpurun_out/r03_n_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['value'], c['stage_ms']['ms_parse'], c['stage_ms']['ms_total'])" || tail -3 gpurun_out/r03_n_$name.log
}
run t128_w2048
run t64_w2048 BROTLI_AMD_TILE_KB=64
run t128_w1024 BROTLI_AMD_TILE_WARM=1024
run t128_w512 BROTLI_AMD_TILE_WARM=512
run t64_w1024 BROTLI_AMD_TILE_KB=64 BROTLI_AMD_TILE_WARM=1024
