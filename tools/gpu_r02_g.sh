#!/bin/bash
# Round 2, GPU session G: wave-level prefix codes (k_prefix.h) in k_store / k_fast_store, restated histogram smoothing.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -m gpu ) > gpurun_out/g_pytest.log 2>&1
tail -3 gpurun_out/g_pytest.log
grep -q " passed" gpurun_out/g_pytest.log && ! grep -q "failed\|Aborted" gpurun_out/g_pytest.log || { echo PARITY FAILED; tail -40 gpurun_out/g_pytest.log | cut -c1-300; exit 1; }
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/g_$name.log 2>&1
  grep "^{" gpurun_out/g_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['ratio'], d['config']['stage_ms'])" || tail -3 gpurun_out/g_$name.log
}
run sw4 ""
run sw3 "" BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_sw3.so
run sw2 "" BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_sw2.so
run q1_text "--quality 1 --data text --lgwin 18"
run q1_random "--quality 1 --data random"
( WHICH=s BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_sprof.so PROBE_SHARDS=131072 timeout 300 python tools/gpu_prof_stages.py ) > gpurun_out/g_store_phases.log 2>&1; grep -A8 STAGE gpurun_out/g_store_phases.log
( WHICH=b BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_bprof.so PROBE_SHARDS=131072 timeout 300 python tools/gpu_prof_stages.py ) > gpurun_out/g_build_phases.log 2>&1; grep -A8 STAGE gpurun_out/g_build_phases.log
