"""Stage timing of one library variant (BROTLI_AMD_HIP_LIB) on a cached input."""
import os, sys, time
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
tag = os.environ.get("VARIANT", "base")
for shard in [int(x) for x in os.environ.get("PROBE_SHARDS", "262144,65536").split(",")]:
    p = hip.make_params(5, 22, shard)
    d_out = torch.empty(ctx.max_output(N, p), dtype=torch.uint8, device="cuda:0")
    for rep in range(2):
        n, info = ctx.encode_device(d_in, N, p, d_out)
    print("VARIANT %-6s shard=%-7d out=%d total=%.1f init=%.1f parse=%.1f build=%.1f store=%.1f -> %.0f MB/s" % (
        tag, shard, n, info["ms_total"], info["ms_init"], info["ms_parse"], info["ms_build"], info["ms_store"],
        N / 1e6 / (info["ms_total"] / 1e3)), flush=True)
    del d_out
