#!/bin/bash
mkdir -p gpurun_out
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/ixe_$name.log 2>&1
  tail -1 gpurun_out/ixe_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['stage_ms'])" || tail -3 gpurun_out/ixe_$name.log
}
run full_cg2 "" BROTLI_AMD_CGROUPS=2
run half_cg4_1024waves "--size-mb 512" BROTLI_AMD_CGROUPS=4
run quarter_cg2_1024waves "--size-mb 256" BROTLI_AMD_CGROUPS=2
run half_cg2_2048waves "--size-mb 512" BROTLI_AMD_CGROUPS=2
run quarter_cg4_512waves "--size-mb 256" BROTLI_AMD_CGROUPS=4
rm -rf gpurun_out/ixe_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/ixe_prof -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/ixe_prof.log 2>&1
python tools/pmc_summary.py gpurun_out/ixe_prof 2>&1 | grep -E "KERNEL k_ix"
find gpurun_out -name "*.db" -delete
