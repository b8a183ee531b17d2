#!/bin/bash
# Round 2, GPU session A: parity, baseline bench of the indexed parse, variants, kernel stats, chain phase timers.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/a_pytest_parity.log 2>&1
tail -3 gpurun_out/a_pytest_parity.log
( time timeout 900 python bench.py ) > gpurun_out/a_bench.log 2>&1
tail -1 gpurun_out/a_bench.log | cut -c1-2500
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/a_$name.log 2>&1
  tail -1 gpurun_out/a_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['ratio'], d['config']['stage_ms'])" || tail -3 gpurun_out/a_$name.log
}
run table "" BROTLI_AMD_INDEXED=0
run narrow_cg2 "" BROTLI_AMD_WIDE=0 BROTLI_AMD_CGROUPS=2
run wide_256k "--shard-kb 256"
run wide_512k "--shard-kb 512"
run wide_1024k "--shard-kb 1024"
run wide_64k "--shard-kb 64"
rm -rf gpurun_out/a_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/a_prof -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/a_prof.log 2>&1
python tools/pmc_summary.py gpurun_out/a_prof > gpurun_out/a_prof_summary.txt 2>&1
grep -E "KERNEL k_" gpurun_out/a_prof_summary.txt
find gpurun_out -name "*.db" -delete
( BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_prof.so PROBE_CHAIN=1 PROBE_MB=1024 PROBE_SHARDS=131072,524288 timeout 600 python tools/gpu_prof_phases.py ) > gpurun_out/a_phases.log 2>&1
cat gpurun_out/a_phases.log | tail -40
