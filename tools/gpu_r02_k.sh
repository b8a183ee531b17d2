#!/bin/bash
mkdir -p gpurun_out
for v in nores ressorted; do
  TAG=$v BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so PROBE_SHARDS=131072 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY
done | tee gpurun_out/k_ix_variants2.log
