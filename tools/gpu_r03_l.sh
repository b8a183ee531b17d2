#!/bin/bash
# round 3, GPU session l: one-shot calls after the fixes (every command under its own timeout), index attribution
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_abi.py -x -q -m gpu -k "one_shot" 2>&1 | tail -2
for v in nowin nores nolong; do
  TAG=$v BROTLI_AMD_HIP_LIB=$PWD/build/var/lib_$v.so timeout 200 python tools/gpu_ix_only.py 2>&1 | grep IXONLY
done
TAG=base timeout 200 python tools/gpu_ix_only.py 2>&1 | grep IXONLY
