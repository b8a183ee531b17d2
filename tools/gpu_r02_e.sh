#!/bin/bash
# Round 2, GPU session E: index with 4-byte entries + 16-byte LDS compares, buckets per wave sweep.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/e_pytest_parity.log 2>&1
tail -3 gpurun_out/e_pytest_parity.log
grep -q " passed" gpurun_out/e_pytest_parity.log && ! grep -q "failed\|Aborted" gpurun_out/e_pytest_parity.log || { echo PARITY FAILED; tail -40 gpurun_out/e_pytest_parity.log | cut -c1-300; exit 1; }
for b in 1 2 4 8; do
  TAG=bpw$b BROTLI_AMD_IX_BPW=$b PROBE_SHARDS=131072,524288 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY
done | tee gpurun_out/e_ix_bpw.log
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/e_$name.log 2>&1
  tail -1 gpurun_out/e_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['ratio'], d['config']['stage_ms'])" || tail -3 gpurun_out/e_$name.log
}
run cg4 "" BROTLI_AMD_CGROUPS=4
run cg4_256k "--shard-kb 256" BROTLI_AMD_CGROUPS=4
run cg4_512k "--shard-kb 512" BROTLI_AMD_CGROUPS=4
rm -rf gpurun_out/e_prof
( cd /tmp && BROTLI_AMD_CGROUPS=4 timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/e_prof -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/e_prof.log 2>&1
python tools/pmc_summary.py gpurun_out/e_prof > gpurun_out/e_prof_summary.txt 2>&1
grep -E "KERNEL k_|KERNEL void" gpurun_out/e_prof_summary.txt
find gpurun_out -name "*.db" -delete
