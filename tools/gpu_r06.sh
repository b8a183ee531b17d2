#!/bin/bash
# tools/gpu_r06.sh SESSION ...: round-6 GPU sessions (run through gpurun from the repo root; results under gpurun_out/r06_SESSION).
#   pmc TAG KERNEL_REGEX [bench args ...]   SQ counters (two passes) + FETCH_SIZE / WRITE_SIZE (a pass each) of
#                                           `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-legs <bench args>`,
#                                           summarised for the kernels matching KERNEL_REGEX (BROTLI_AMD_HIP_LIB selects a variant)
#   stats TAG [bench args ...]              rocprofv3 --kernel-trace --stats of the same command (steps 3)
s=$1; shift
export TMPDIR=/tmp
case $s in
  pmc)
    tag=$1; re=$2; shift 2
    out=gpurun_out/r06_pmc_$tag; mkdir -p $out; i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
               "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
               "FETCH_SIZE" "WRITE_SIZE" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
      i=$((i+1))
      ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$out/p$i -o p$i -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-legs "$@" ) > $out/p$i.log 2>&1
    done
    python tools/pmc_summary.py $out | grep -E "^DB|$re" > $out/summary.txt
    find $out -name "*.db" -delete
    cat $out/summary.txt ;;
  stats)
    tag=$1; shift
    out=gpurun_out/r06_stats_$tag; mkdir -p $out
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$out/s -o s -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs "$@" ) > $out/bench.log 2>&1
    python tools/pmc_summary.py $out | grep -E "KERNEL" | sort -t= -k3 -n -r > $out/kernel_stats.txt
    find $out -name "*.db" -delete
    head -30 $out/kernel_stats.txt ;;
esac
