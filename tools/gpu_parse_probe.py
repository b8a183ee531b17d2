"""GPU probe: parse-only parity vs the oracle command tap + timing sweep."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gen_inputs as G
from brotli_amd import hip
from refharness import Oracle
from simharness import oracle_commands

ctx = hip.Context(0)
o = Oracle()
res = []
# parity
PARITY = [] if os.environ.get("PROBE_SKIP_PARITY") else None
for name, data, hint, shard in [
        ("alice", open(os.path.join(ROOT, "tests/golden/alice29.txt"), "rb").read(), 0, 0),
        ("text4m/256k", G.enwik_text(4 << 20, seed=11, vocab=20000), 1 << 30, 1 << 18),
        ("mixed2m/128k", G.mixed_corpus(2 << 20), 1 << 30, 1 << 17),
        ("text_rand", G.enwik_text(200000, seed=3) + G.random_bytes(150000) + G.enwik_text(100000, seed=4), 1 << 30, 0)][:0 if PARITY is not None else 9]:
    want = oracle_commands(o, data, 5, 22, hint, shard)
    d = hip.to_device(data)
    got, info = ctx.debug_parse(d, len(data), hip.make_params(5, 22, shard, hint))
    ok = bool(np.array_equal(want, got))
    print("PARITY", name, len(want), len(got), ok, info, flush=True)
    res.append({"case": name, "ok": ok})
# timing sweep
N = int(os.environ.get("PROBE_MB", "256")) << 20
t0 = time.time(); data = G.enwik_text(N); print("gen %.1fs" % (time.time() - t0), flush=True)
d = hip.to_device(data)
SHARDS = [int(x) for x in os.environ.get("PROBE_SHARDS", "1048576,524288,262144,131072,65536").split(",")]
REPS = int(os.environ.get("PROBE_REPS", "2"))
for shard in SHARDS:
    for rep in range(REPS):
        got, info = ctx.debug_parse(d, N, hip.make_params(5, 22, shard, 1 << 30))
    mbps = N / 1e6 / (info["ms_parse"] / 1e3)
    print("SWEEP shard=%d nshards=%d parse=%.2fms init=%.2fms -> %.0f MB/s searches/B=%.3f steps/B=%.3f cmds=%d" % (
        shard, info["nshards"], info["ms_parse"], info["ms_init"], mbps, info["searches"] / N, info["search_steps"] / N, info["commands"]), flush=True)
    res.append({"shard": shard, **info, "MBps": mbps})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "parse_probe.json"), "w"), indent=1)
