#!/bin/bash
# tools/gpu_r05.sh SESSION [variant.so ...]: round-5 GPU sessions (run through gpurun from the repo root).
#   ix  : the index kernels alone (tools/gpu_ix_only.py) for the shipped library and every variant named, text and mix,
#         128 KiB and 1 MiB shards
#   bench: the bench line of the shipped library (no legs, no baseline)
s=$1; shift
out=gpurun_out/r05_$s; mkdir -p $out
case $s in
  ix*)
    for lib in brotli_amd/lib/libbrotli_amd_hip.so "$@"; do
      for kind in text mix; do
        TAG=$(basename $lib .so) BROTLI_AMD_HIP_LIB=$PWD/$lib PROBE_KIND=$kind PROBE_SHARDS=131072,1048576 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY
      done
    done | tee $out/ix_only.txt ;;
  bench*)
    for lib in brotli_amd/lib/libbrotli_amd_hip.so "$@"; do
      BROTLI_AMD_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$(basename $lib .so).json 2>$out/bench.err
      python - <<P
import json
d=json.loads([l for l in open("$out/bench_$(basename $lib .so).json") if l.startswith("{")][-1])
print("$lib", d["value"], d["config"]["stage_ms"], d["config"]["compressed_bytes"])
P
    done | tee $out/bench.txt ;;
esac
# (appended) pmc SESSION lib...: SQ counters of the index / chain / build / store kernels for each library named
if [ "${s#pmc}" != "$s" ]; then
  export TMPDIR=/tmp
  for lib in "$@"; do
    tag=$(basename $lib .so); i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY" \
               "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
      i=$((i+1))
      ( cd /tmp && BROTLI_AMD_HIP_LIB=/root/repo/$lib timeout 300 rocprofv3 --pmc $set --kernel-trace -d /root/repo/$out/$tag/p$i -o p$i -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} ) > $out/${tag}_p$i.log 2>&1
    done
    python tools/pmc_summary.py $out/$tag | grep -E "^DB|k_ix_bucket|k_chain|k_build|k_store" > $out/${tag}_summary.txt
    find $out/$tag -name "*.db" -delete
  done
  tail -n 60 $out/*_summary.txt
fi
