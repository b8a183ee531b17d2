#!/bin/bash
# First MI355X session of the indexed parse: parity tests, then bench with and without the index.
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > gpurun_out/ix_pytest_parity.log 2>&1
tail -5 gpurun_out/ix_pytest_parity.log
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/ix_bench_indexed.log 2>&1; tail -1 gpurun_out/ix_bench_indexed.log | cut -c1-1200
( BROTLI_AMD_INDEXED=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/ix_bench_table.log 2>&1; tail -1 gpurun_out/ix_bench_table.log | cut -c1-1200
for g in 1 2 4; do
( BROTLI_AMD_CGROUPS=$g timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/ix_bench_cg$g.log 2>&1; echo cgroups $g; tail -1 gpurun_out/ix_bench_cg$g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['stage_ms'])"
done
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --shard-kb 512 ) > gpurun_out/ix_bench_512k.log 2>&1; tail -1 gpurun_out/ix_bench_512k.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(512, d['value'], d['config']['stage_ms'], d['config']['ratio'])"
rm -rf gpurun_out/ix_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/ix_prof -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/ix_prof.log 2>&1
python tools/pmc_summary.py gpurun_out/ix_prof > gpurun_out/ix_prof_summary.txt 2>&1
grep -E "KERNEL k_" gpurun_out/ix_prof_summary.txt
find gpurun_out -name "*.db" -delete
