#!/bin/bash
# tools/gpu_ws.sh LIB BPW...: WRITE_SIZE / FETCH_SIZE and duration of the index kernels of LIB at the given buckets per wave
lib=$1; shift
export TMPDIR=/tmp
out=gpurun_out/r05_ws; mkdir -p $out
for bpw in "$@"; do
  for c in WRITE_SIZE FETCH_SIZE; do
    ( cd /tmp && BROTLI_AMD_IX_BPW=$bpw BROTLI_AMD_HIP_LIB=/root/repo/$lib timeout 300 rocprofv3 --pmc $c --kernel-trace -d /root/repo/$out/b${bpw}_$c -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} ) > $out/b${bpw}_$c.log 2>&1
    echo "== bpw=$bpw $c"; python tools/pmc_summary.py $out/b${bpw}_$c | grep -E "k_ix_bucket|k_ix_scatter|k_chain" | grep -E "PMC|KERNEL"
  done
done 2>&1 | tee $out/summary_$(basename $lib .so).txt
find $out -name "*.db" -delete
