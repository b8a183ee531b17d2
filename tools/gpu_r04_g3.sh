#!/bin/bash
# Round 4, session g3: the stock calls with the caller's buffers used directly; quality 9 with more waves in flight
# (smaller shards; k_parse_deep compiled for four waves per SIMD).
ulimit -c 0
O=gpurun_out/r04g3
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tools/stock_call.py 1024 22 text 4 --ref > $O/stock_1024.log 2>&1
echo "stock 1 GiB rc $?: $(grep '"stage": "done"' $O/stock_1024.log | tail -1)" | tee $O/summary.txt
timeout 200 python tools/stock_call.py 3.99 22 text 5 --ref > $O/stock_4.log 2>&1
echo "stock 4 MiB rc $?: $(grep '"stage": "done"' $O/stock_4.log | tail -1)" | tee -a $O/summary.txt
timeout 200 python tools/stock_call.py 64 22 text 4 --ref > $O/stock_64.log 2>&1
echo "stock 64 MiB rc $?: $(grep '"stage": "done"' $O/stock_64.log | tail -1)" | tee -a $O/summary.txt
echo "== quality 9, lgwin 24" | tee -a $O/summary.txt
for kb in 512 384 256; do
  timeout 300 python bench.py --quality 9 --lgwin 24 --shard-kb $kb --steps 2 --no-cpu-baseline > $O/bench_q9_${kb}.json 2> $O/bench_q9_${kb}.err
  echo "q9 ${kb} KiB rc $?" | tee -a $O/summary.txt
done
BROTLI_AMD_HIP_LIB=$PWD/build/var/deepw4.so timeout 300 python bench.py --quality 9 --lgwin 24 --shard-kb 256 --steps 2 --no-cpu-baseline > $O/bench_q9_256_w4.json 2> $O/bench_q9_256_w4.err
echo "q9 256 KiB, four waves per SIMD rc $?" | tee -a $O/summary.txt
BROTLI_AMD_HIP_LIB=$PWD/build/var/deepw4.so timeout 300 python bench.py --quality 9 --lgwin 24 --shard-kb 128 --steps 2 --no-cpu-baseline > $O/bench_q9_128_w4.json 2> $O/bench_q9_128_w4.err
echo "q9 128 KiB, four waves per SIMD rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g3/bench*.json")):
    try:
        d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], "ratio", d["config"].get("ratio"), d["config"].get("stage_ms"), d["config"].get("device_round_trip", {}).get("equal_to_input"))
    except Exception as e:
        print(f, "unreadable", e)
PY
