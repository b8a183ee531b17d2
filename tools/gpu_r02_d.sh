#!/bin/bash
# Round 2, GPU session D: where the index time goes (variants of k_ix_bucket, index kernels only).
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in base bpw4 bpw2 bpw1 nowin nolong; do
  TAG=$v BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so PROBE_SHARDS=131072,524288 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY
done | tee gpurun_out/d_ix_variants.log
