"""Times the index kernels alone (BROTLI_AMD_INDEX_ONLY=1) for the library in BROTLI_AMD_HIP_LIB.
PROBE_SHARDS: shard sizes, PROBE_BPW: values of BROTLI_AMD_IX_BPW to try (0 = the planner's choice), PROBE_KIND: text / mix."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BROTLI_AMD_INDEX_ONLY"] = "1"
import torch
import gen_inputs as G
from brotli_amd import hip
N = int(os.environ.get("PROBE_MB", "1024")) << 20
kind = os.environ.get("PROBE_KIND", "text")
data = G.enwik_text(N) if kind == "text" else G.mixed_corpus(N)
ctx = hip.Context(0)
d = hip.to_device(data)
for shard in [int(x) for x in os.environ.get("PROBE_SHARDS", "131072").split(",")]:
    for bpw in [int(x) for x in os.environ.get("PROBE_BPW", "0").split(",")]:
        if bpw:
            os.environ["BROTLI_AMD_IX_BPW"] = str(bpw)
        else:
            os.environ.pop("BROTLI_AMD_IX_BPW", None)
        hip.refresh_env()
        ms, msb = [], []
        for rep in range(3):
            got, info = ctx.debug_parse(d, N, hip.make_params(5, 22, shard, 1 << 30))
            ms.append(info["ms_index"])
            msb.append(info.get("ms_ix_bucket", 0.0))
        print("IXONLY %s %s shard=%d bpw=%d ms_index=%s ms_ix_bucket=%s" % (
            os.environ.get("TAG", ""), kind, shard, bpw, " ".join("%.2f" % m for m in ms), " ".join("%.2f" % m for m in msb)), flush=True)
