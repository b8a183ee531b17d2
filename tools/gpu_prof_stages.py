"""Phase cycle breakdown of k_store / k_build (libraries built with -DS_PROFILE / -DB_PROFILE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gen_inputs as G
from brotli_amd import hip
N = int(os.environ.get("PROBE_MB", "1024")) << 20
cache = "/dev/shm/brotli_amd_probe_%d.bin" % N
if os.path.exists(cache):
    data = open(cache, "rb").read()
else:
    data = G.enwik_text(N); open(cache, "wb").write(data)
ctx = hip.Context(0)
d_in = hip.to_device(data)
which = os.environ.get("WHICH", "s")
names = {"s": ["zero+setup", "small histos", "tree jobs", "switch codes", "header", "commands", "-", "-"],
         "b": ["compress?+contexts", "streams", "split lit", "split cmd", "split dist", "rle smoothing", "-", "-"]}[which]
for shard in [int(x) for x in os.environ.get("PROBE_SHARDS", "262144,65536").split(",")]:
    p = hip.make_params(5, 22, shard)
    d_out = torch.empty(ctx.max_output(N, p), dtype=torch.uint8, device="cuda:0")
    for rep in range(2):
        n, info = ctx.encode_device(d_in, N, p, d_out)
    prof = info["prof"]; tot = sum(prof) or 1
    print("STAGE %s shard=%d build=%.1fms store=%.1fms cycles/shard=%.0f" % (which, shard, info["ms_build"], info["ms_store"], tot / info["nshards"]))
    for nme, pv in zip(names, prof):
        if pv: print("   %-18s %5.1f%%  %.0f cycles per shard" % (nme, 100.0 * pv / tot, pv / info["nshards"]))
    del d_out
