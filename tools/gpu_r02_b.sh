#!/bin/bash
# Round 2, GPU session B: group fast path of k_chain (taint bits in res[]), shards-per-wave sweep, phases, index PMC.
mkdir -p gpurun_out
export TMPDIR=/tmp
./build/dpp_probe
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/b_pytest_parity.log 2>&1
tail -3 gpurun_out/b_pytest_parity.log
grep -q " passed" gpurun_out/b_pytest_parity.log && ! grep -q "failed\|Aborted" gpurun_out/b_pytest_parity.log || { echo PARITY FAILED; tail -40 gpurun_out/b_pytest_parity.log | cut -c1-300; exit 1; }
run() {  # name, args, env...
  local name=$1; local args=$2; shift; shift
  ( env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > gpurun_out/b_$name.log 2>&1
  tail -1 gpurun_out/b_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['config']['ratio'], d['config']['stage_ms'])" || tail -3 gpurun_out/b_$name.log
}
run cg_default ""
run cg1 "" BROTLI_AMD_CGROUPS=1
run cg2 "" BROTLI_AMD_CGROUPS=2
run cg4 "" BROTLI_AMD_CGROUPS=4
run wide "" BROTLI_AMD_WIDE=1
run cg4_256k "--shard-kb 256" BROTLI_AMD_CGROUPS=4
run cg1_256k "--shard-kb 256" BROTLI_AMD_CGROUPS=1
run cg4_512k "--shard-kb 512" BROTLI_AMD_CGROUPS=4
run cg1_512k "--shard-kb 512" BROTLI_AMD_CGROUPS=1
run cg1_1024k "--shard-kb 1024" BROTLI_AMD_CGROUPS=1
run cg4_64k "--shard-kb 64" BROTLI_AMD_CGROUPS=4
( BROTLI_AMD_CGROUPS=4 BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_prof.so PROBE_CHAIN=1 PROBE_MB=1024 PROBE_SHARDS=131072 timeout 600 python tools/gpu_prof_phases.py ) > gpurun_out/b_phases.log 2>&1
tail -14 gpurun_out/b_phases.log
OUT=gpurun_out/b_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$OUT/p$i -o p$i -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ) > $OUT/p$i.log 2>&1
done
python tools/pmc_summary.py $OUT | grep -E "^DB|k_ix|k_chain|k_build|k_store" > $OUT/summary.txt
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
