#!/bin/bash
# Round 4, session f: the big-bucket path of k_ix_bucket with its loads ahead of its rows (1 MiB shards, the mix's
# runs of zeros), the chain's 128-entry rounds in exact searches (the mix), parity on the new code.
ulimit -c 0
O=gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest (parity file + tiles)" | tee $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zx_tiles.py -q -m gpu -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?: $(tail -1 $O/pytest.log)" | tee -a $O/summary.txt
echo "== index kernels alone" | tee -a $O/summary.txt
TAG=G PROBE_SHARDS=131072,1048576 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
TAG=G PROBE_KIND=mix PROBE_SHARDS=131072 timeout 300 python tools/gpu_ix_only.py 2>&1 | grep IXONLY | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
timeout 300 python bench.py --shard-kb 1024 --steps 3 --no-cpu-baseline > $O/bench_1024k.json 2> $O/bench_1024k.err
echo "bench 1 MiB rc $?" | tee -a $O/summary.txt
timeout 400 python bench.py --workload silesia --steps 3 --no-cpu-baseline > $O/bench_mix.json 2> $O/bench_mix.err
echo "bench mix rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04f/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("stage_ms"), d["config"].get("device_round_trip", {}).get("equal_to_input"))
    except Exception as e:
        print(f, "unreadable", e)
PY
