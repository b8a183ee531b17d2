#!/bin/bash
# Round 4, session h (after the final run; nothing of it ships): where the chain's cycles go on the member kinds of the
# mix, each alone (library built with -DQ_PROFILE: cycles per phase, summed over the shards' lane 0).
ulimit -c 0
O=gpurun_out/r04h
mkdir -p $O
export TMPDIR=/tmp
for kind in text zeros gradient floats; do
  PROBE_CHAIN=1 PROBE_KIND=$kind PROBE_MB=256 PROBE_SHARDS=131072 BROTLI_AMD_HIP_LIB=$PWD/build/var/qprof.so timeout 200 python tools/gpu_prof_phases.py 2>&1 | grep -A13 PHASES | tee -a $O/summary.txt
done
