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
# bench.py releases the device path's context before its ABI legs now: the default line and the quality-9 line once more
( time timeout 600 python bench.py --steps 3 ) > $O/bench.json 2> $O/bench.err
echo "bench rc $?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --quality 9 --lgwin 24 --shard-kb 384 --steps 2 ) > $O/bench_q9.json 2> $O/bench_q9.err
echo "q9 rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r04h/bench*.json")):
    try:
        d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith("{")][-1])
        c = d["config"]; b = d.get("cpu_baseline") or {}
        print(f.split("/")[-1], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms | ratio", c.get("ratio"), "| cpu", b.get("value"), "| sha", c.get("parity_full_sha256_equal"),
              "| traffic", d["roofline"].get("traffic"), "| abi", c.get("end_to_end_abi"), "| stock", (c.get("stock_call_no_plan") or {}).get("whole_input"))
    except Exception as e:
        print(f, "unreadable", e)
PY
