"""tools/cpu_scaling_probe.py [MiB]: bench.py's cpu_baseline (the scaling study included) on this box's host cores, alone
(no GPU work): the JSON object bench.py would put under `cpu_baseline`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import gen_inputs as G
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
t0 = time.time()
data = G.enwik_text(mb << 20, seed=G.SEED)
path = bench.data_file(data)
t1 = time.time()
try:
    cb = bench.cpu_baseline(path, len(data), 5, 22, 128 << 10, min(len(data), 1 << 30), reps=5, other_plans=[1 << 20])
finally:
    os.unlink(path)
cb["probe_seconds"] = {"generate": round(t1 - t0, 1), "baseline": round(time.time() - t1, 1)}
print(json.dumps(cb))
