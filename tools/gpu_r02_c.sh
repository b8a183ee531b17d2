#!/bin/bash
# Round 2, GPU session C: lean commit (raw commands + k_cmd_encode), branch-free evaluation, dictionary probe in the fast path.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/c_pytest_parity.log 2>&1
tail -3 gpurun_out/c_pytest_parity.log
grep -q " passed" gpurun_out/c_pytest_parity.log && ! grep -q "failed\|Aborted" gpurun_out/c_pytest_parity.log || { echo PARITY FAILED; tail -40 gpurun_out/c_pytest_parity.log | cut -c1-300; exit 1; }
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/c_$name.log 2>&1
  tail -1 gpurun_out/c_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['ratio'], d['config']['stage_ms'])" || tail -3 gpurun_out/c_$name.log
}
run cg2 "" BROTLI_AMD_CGROUPS=2
run cg4 "" BROTLI_AMD_CGROUPS=4
run cg4_cw2 "" BROTLI_AMD_CGROUPS=4 BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_cw2.so
run cg4_256k "--shard-kb 256" BROTLI_AMD_CGROUPS=4
run cg4_512k "--shard-kb 512" BROTLI_AMD_CGROUPS=4
run cg4_64k "--shard-kb 64" BROTLI_AMD_CGROUPS=4
( BROTLI_AMD_CGROUPS=4 BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_prof.so PROBE_CHAIN=1 PROBE_MB=1024 PROBE_SHARDS=131072 timeout 600 python tools/gpu_prof_phases.py ) > gpurun_out/c_phases.log 2>&1
tail -14 gpurun_out/c_phases.log
