"""Phase cycle breakdown of k_parse4 (library built with -DQ_PROFILE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gen_inputs as G
from brotli_amd import hip
N = int(os.environ.get("PROBE_MB", "256")) << 20
KIND = os.environ.get("PROBE_KIND", "text")      # text / mix / zeros / floats / gradient / noise: one member kind of the mix alone
if KIND == "text":
    data = G.enwik_text(N)
elif KIND == "mix":
    data = G.mixed_corpus(N)
else:
    rng = np.random.default_rng(G.SEED)
    if KIND == "zeros":
        z = np.zeros(N, dtype=np.uint8)
        pos = rng.integers(0, N, size=N // 50)
        z[pos] = rng.integers(1, 256, size=pos.size)
        data = z.tobytes()
    elif KIND == "floats":
        data = np.cumsum(rng.normal(size=N // 4)).astype(np.float32).tobytes()
    elif KIND == "rows":
        a = rng.integers(0, 10 ** 6, size=N // 18 + 1)
        data = b"".join(b"%08d|%06d|OK\n" % (i, v) for i, v in enumerate(a))[:N]
    elif KIND == "gradient":
        data = ((np.arange(N) // 7 + rng.integers(0, 3, size=N)) & 255).astype(np.uint8).tobytes()
    else:
        data = G.random_bytes(N, G.SEED)
ctx = hip.Context(0)
d = hip.to_device(data)
QUALITY, LGWIN = int(os.environ.get("PROBE_QUALITY", "5")), int(os.environ.get("PROBE_LGWIN", "22"))
if QUALITY >= 6:
    names = ["bytes at P -> counter + record", "candidate strings", "scores + reductions", "step-by-step resolve", "insert", "dictionaries", "decide + commit", "StoreRange", "driver", "-", "-", "-"]
elif os.environ.get("PROBE_CHAIN"):
    names = ["top/driver/marks", "loads -> dc len", "eval rest + bcast", "lean loop", "sr fetch + commit", "generic", "accounting", "post", "FAST loads+eval", "FAST exact fix", "FAST group logic", "FAST commit+marks"]
else:
  names = ["record(after P bytes)", "ext+rest of cand", "resolve", "insert", "dict", "decide", "stores", "driver", "bytes at P", "dc strings", "bucket strings", "-"]
for shard in [int(x) for x in os.environ.get("PROBE_SHARDS", "262144,65536").split(",")]:
    for rep in range(2):
        got, info = ctx.debug_parse(d, N, hip.make_params(QUALITY, LGWIN, shard, 1 << 30))
    prof = info["prof"]; tot = sum(prof)
    iters = info["search_steps"]
    print("PHASES kind=" + KIND + " shard=%d parse=%.1fms searches=%d cmds=%d total_cycles/shard-iter=%.0f" % (
        shard, info["ms_parse"], info["searches"], info["commands"], tot / max(1, iters)))
    for n, p in zip(names, prof):
        print("   %-10s %5.1f%%  %.0f cycles per search" % (n, 100.0 * p / tot, p / max(1, iters)))
