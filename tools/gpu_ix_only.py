"""Times the index kernels alone (BROTLI_AMD_INDEX_ONLY=1) for the library in BROTLI_AMD_HIP_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BROTLI_AMD_INDEX_ONLY"] = "1"
import torch
import gen_inputs as G
from brotli_amd import hip
N = int(os.environ.get("PROBE_MB", "1024")) << 20
data = G.enwik_text(N)
ctx = hip.Context(0)
d = hip.to_device(data)
for shard in [int(x) for x in os.environ.get("PROBE_SHARDS", "131072").split(",")]:
    ms = []
    for rep in range(3):
        got, info = ctx.debug_parse(d, N, hip.make_params(5, 22, shard, 1 << 30))
        ms.append(info["ms_index"])
    print("IXONLY %s shard=%d ms_index=%s" % (os.environ.get("TAG", ""), shard, " ".join("%.2f" % m for m in ms)), flush=True)
