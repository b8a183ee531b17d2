#!/bin/bash
# bench stage times of the indexed parse for several library variants (build/var/lib_*.so) and wave layouts
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} ) > gpurun_out/ixv_$name.log 2>&1
  tail -1 gpurun_out/ixv_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['stage_ms'])" || tail -3 gpurun_out/ixv_$name.log
}
run base_cg2 BROTLI_AMD_CGROUPS=2
run base_cg4 BROTLI_AMD_CGROUPS=4
for v in pd32 noearly w2 w3; do
  run ${v}_cg2 BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so BROTLI_AMD_CGROUPS=2
done
run w2_cg4 BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_w2.so BROTLI_AMD_CGROUPS=4
run pd32_cg4 BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_pd32.so BROTLI_AMD_CGROUPS=4
