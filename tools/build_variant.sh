#!/bin/bash
# tools/build_variant.sh NAME [SRC_ROOT] [-Dflags...]: an experiment build of the HIP library into build/var/NAME.so
# (selected at run time with BROTLI_AMD_HIP_LIB; build/ is git-ignored but travels to the GPU box).
name=$1; shift
root=/root/repo
if [ -d "$1" ]; then root=$1; shift; fi
mkdir -p /root/repo/build/var
/opt/rocm/bin/hipcc -O3 -ffp-contract=off -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" \
  -I$root/include -shared $root/brotli_amd/csrc/hip_layer.hip -lpthread -o /root/repo/build/var/$name.so 2>&1 | grep -E "error" 
echo "built build/var/$name.so"
