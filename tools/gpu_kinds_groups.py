"""tools/gpu_kinds_groups.py: the chain's time per member kind of the mix at FULL width (1 GiB = 8192 shards of 128 KiB:
64 MiB of the kind, sixteen times) with 4 and with 2 shards per wave (BROTLI_AMD_CGROUPS) — which kinds the lock-step of
four 16-lane groups costs (the mix as a whole: 134 ms at 4, 109 ms at 2; text: 24 against 39)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_inputs as G
from brotli_amd import hip
NAMES = ["text", "xml", "source", "rows", "floats", "gradients", "sparse zeros", "noise"]


def one_kind(t, n):
    """n bytes of member kind t of tests/gen_inputs.mixed_corpus (tools/gpu_mix_kinds.py)."""
    out = bytearray()
    k = t
    while len(out) < n:
        piece = G.mixed_corpus(12 << 20, seed=G.SEED + 17 * k)
        out += piece[t << 20:(t + 1) << 20] if t < 7 else piece[7 << 20:(7 << 20) + (1 << 18)]
        k += 8
    return bytes(out[:n])


BASE = int(os.environ.get("PROBE_BASE_MB", "64")) << 20
REP = int(os.environ.get("PROBE_REP", "16"))
ctx = None
for t, name in enumerate(NAMES):
    data = one_kind(t, BASE) * REP
    n = len(data)
    line = "KIND %-13s" % name
    for g in (4, 2):
        os.environ["BROTLI_AMD_CGROUPS"] = str(g)
        hip.refresh_env()
        if ctx is None:
            ctx = hip.Context(0)
        d = hip.to_device(data)
        best = None
        for rep in range(2):
            got, info = ctx.debug_parse(d, n, hip.make_params(5, 22, 131072, 1 << 30))
            best = info if best is None or info["ms_parse"] < best["ms_parse"] else best
        line += "  groups %d: parse=%7.1f ms" % (g, best["ms_parse"])
        del d
    print(line + "  exact=%d searches=%d" % (best.get("exact_searches", 0), best["searches"]), flush=True)
