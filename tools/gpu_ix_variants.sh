for lib in "$@"; do TAG=$(basename $lib .so) BROTLI_AMD_HIP_LIB=$PWD/$lib PROBE_KIND=text PROBE_SHARDS=131072 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY; done
