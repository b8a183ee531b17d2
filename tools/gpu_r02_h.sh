#!/bin/bash
# Round 2, GPU session H: LDS-staged first-level scatter of the index.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/h_pytest.log 2>&1
tail -3 gpurun_out/h_pytest.log
grep -q " passed" gpurun_out/h_pytest.log && ! grep -q "failed\|Aborted" gpurun_out/h_pytest.log || { echo PARITY FAILED; tail -40 gpurun_out/h_pytest.log | cut -c1-300; exit 1; }
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/h_$name.log 2>&1
  grep "^{" gpurun_out/h_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['ratio'], d['config']['stage_ms'])" || tail -3 gpurun_out/h_$name.log
}
run base ""
run s256k "--shard-kb 256"
run s512k "--shard-kb 512"
rm -rf gpurun_out/h_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/h_prof -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/h_prof.log 2>&1
python tools/pmc_summary.py gpurun_out/h_prof > gpurun_out/h_prof_summary.txt 2>&1
grep -E "KERNEL k_" gpurun_out/h_prof_summary.txt
find gpurun_out -name "*.db" -delete
