#!/bin/bash
mkdir -p gpurun_out
VARIANT=base python tools/gpu_variant_probe.py 2>&1 | grep VARIANT
for f in build/var/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  BROTLI_AMD_HIP_LIB=$PWD/$f VARIANT=$n python tools/gpu_variant_probe.py 2>&1 | grep -E "VARIANT|Error|error" | head -5
done
