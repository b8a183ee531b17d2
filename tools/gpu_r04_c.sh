#!/bin/bash
# Round 4, session c: the many-wave build / store (k_wide.h) and the filter-word window search of k_ix_bucket on the
# MI355X: the whole GPU suite, stock calls with stage times, index variants timed alone + their HBM write / fetch
# counters, bench lines at 128 KiB and 1 MiB shards.
ulimit -c 0
O=gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $O/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $O/pytest.log)" | tee -a $O/summary.txt
echo "== stock calls" | tee -a $O/summary.txt
BROTLI_AMD_TILE_LOG=1 timeout 400 python tools/stock_call.py 1024 22 text 3 --ref > $O/stock_1024.log 2>&1
echo "stock 1 GiB rc $?: $(grep '"stage": "done"' $O/stock_1024.log | tail -1)" | tee -a $O/summary.txt
BROTLI_AMD_TILE_LOG=1 BROTLI_AMD_WIDE=0 timeout 400 python tools/stock_call.py 1024 22 text 2 > $O/stock_1024_narrow.log 2>&1
echo "stock 1 GiB, one wave per meta-block rc $?: $(grep '"stage": "done"' $O/stock_1024_narrow.log | tail -1)" | tee -a $O/summary.txt
timeout 200 python tools/stock_call.py 3.99 22 text 5 --ref > $O/stock_4.log 2>&1
echo "stock 4 MiB rc $?: $(grep '"stage": "done"' $O/stock_4.log | tail -1)" | tee -a $O/summary.txt
BROTLI_AMD_WIDE=0 timeout 200 python tools/stock_call.py 3.99 22 text 5 > $O/stock_4_narrow.log 2>&1
echo "stock 4 MiB, one wave rc $?: $(grep '"stage": "done"' $O/stock_4_narrow.log | tail -1)" | tee -a $O/summary.txt
timeout 200 python tools/stock_call.py 32 22 text 4 --ref > $O/stock_32.log 2>&1
echo "stock 32 MiB rc $?: $(grep '"stage": "done"' $O/stock_32.log | tail -1)" | tee -a $O/summary.txt
BROTLI_AMD_TILE_LOG=1 timeout 300 python tools/stock_call.py 64 22 mix 1 --ref > $O/stock_64_mix.log 2>&1
echo "stock 64 MiB mix rc $?: $(grep '"stage": "done"' $O/stock_64_mix.log | tail -1)" | tee -a $O/summary.txt
echo "== index variants (k_ix_* alone, 1 GiB text)" | tee -a $O/summary.txt
for v in ixbase ixfilter ixfilter_nt; do
  TAG=$v BROTLI_AMD_HIP_LIB=$PWD/build/var/$v.so PROBE_SHARDS=131072,1048576 timeout 200 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
done
for v in ixfilter; do
  for pmc in WRITE_SIZE FETCH_SIZE; do
    ( cd /tmp && TAG=$v BROTLI_AMD_HIP_LIB=/root/repo/build/var/$v.so PROBE_SHARDS=131072 timeout 300 rocprofv3 --pmc $pmc --kernel-trace -d /root/repo/$O/pmc_${v}_$pmc -o p -- python /root/repo/tools/gpu_ix_only.py ) > $O/pmc_${v}_$pmc.log 2>&1
  done
done
python tools/pmc_summary.py $O 2>/dev/null | grep -E "^DB|k_ix" > $O/pmc_summary.txt
find $O -name "*.db" -delete
cat $O/pmc_summary.txt | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
timeout 300 python bench.py --shard-kb 1024 --steps 3 --no-cpu-baseline > $O/bench_1024k.json 2> $O/bench_1024k.err
echo "bench 1 MiB shards rc $?" | tee -a $O/summary.txt
BROTLI_AMD_WIDE=0 timeout 300 python bench.py --shard-kb 1024 --steps 3 --no-cpu-baseline > $O/bench_1024k_narrow.json 2> $O/bench_1024k_narrow.err
echo "bench 1 MiB shards, one wave per meta-block rc $?" | tee -a $O/summary.txt
BROTLI_AMD_WIDE=1 timeout 300 python bench.py --steps 3 --no-cpu-baseline > $O/bench_128k_wide.json 2> $O/bench_128k_wide.err
echo "bench 128 KiB shards through the many-wave kernels (K = 4) rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04c/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("stage_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
