#!/bin/bash
# Round 2, GPU session I: res[] / input prefetch in the chain's fast step (A/B), whole-output parity at 1 GiB.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/i_pytest.log 2>&1
tail -3 gpurun_out/i_pytest.log
grep -q " passed" gpurun_out/i_pytest.log && ! grep -q "failed\|Aborted" gpurun_out/i_pytest.log || { echo PARITY FAILED; tail -40 gpurun_out/i_pytest.log | cut -c1-300; exit 1; }
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/i_$name.log 2>&1
  grep "^{" gpurun_out/i_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['ratio'], d['config']['stage_ms'])" || tail -3 gpurun_out/i_$name.log
}
run prefetch ""
run noprefetch "" BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_nopf.so
run prefetch_512k "--shard-kb 512"
run noprefetch_512k "--shard-kb 512" BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_nopf.so
( time timeout 1500 python -m pytest tests/test_gpu_zy_full_size.py -x -q -m gpu ) > gpurun_out/i_pytest_full.log 2>&1
tail -5 gpurun_out/i_pytest_full.log | cut -c1-300
