#!/bin/bash
# round 3, GPU session g: buckets per wave of k_ix_bucket at five waves per SIMD (how many shards an XCD's L2 holds at once)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/r03_g_$name.log 2>&1
  grep '^{' gpurun_out/r03_g_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$name', d['value'], c['ratio'], c['stage_ms'])" || tail -5 gpurun_out/r03_g_$name.log
}
BENCH_ARGS="" run bpw1 BROTLI_AMD_IX_BPW=1
BENCH_ARGS="" run bpw4 BROTLI_AMD_IX_BPW=4
BENCH_ARGS="--shard-kb 1024" run bpw1_1024k BROTLI_AMD_IX_BPW=1
BENCH_ARGS="--shard-kb 1024" run bpw2_1024k BROTLI_AMD_IX_BPW=2
BENCH_ARGS="--shard-kb 1024" run bpw8_1024k BROTLI_AMD_IX_BPW=8
BENCH_ARGS="--shard-kb 256" run tiled_256k
BENCH_ARGS="--shard-kb 512" run tiled_512k
