#!/bin/bash
# Round 4, session g1: the chain with runs tainted as one stretch (text, the mix), 1024 first-level buckets per 128 KiB shard.
ulimit -c 0
O=gpurun_out/r04g1
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest (parity file)" | tee $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $O/pytest.log)" | tee -a $O/summary.txt
echo "== index kernels alone" | tee -a $O/summary.txt
TAG=target320 PROBE_SHARDS=131072 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
TAG=target160 BROTLI_AMD_IX_TARGET=160 PROBE_SHARDS=131072 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
BROTLI_AMD_IX_TARGET=160 timeout 300 python bench.py --steps 5 --no-cpu-baseline > $O/bench_t160.json 2> $O/bench_t160.err
echo "bench (1024 buckets per shard) rc $?" | tee -a $O/summary.txt
timeout 400 python bench.py --workload silesia --steps 3 --no-cpu-baseline > $O/bench_mix.json 2> $O/bench_mix.err
echo "bench mix rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g1/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("stage_ms"), d["config"].get("device_round_trip", {}).get("equal_to_input"))
    except Exception as e:
        print(f, "unreadable", e)
PY
