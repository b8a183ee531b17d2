"""tools/gpu_mix_kinds.py: the chain's time for each member kind of tests/gen_inputs.mixed_corpus alone (PROBE_MB MiB of
one kind, quality 5, lgwin 22, 128 KiB shards): the mix's chain time is its slowest kind's, so this says which."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gen_inputs as G
from brotli_amd import hip
N = int(os.environ.get("PROBE_MB", "256")) << 20
NAMES = ["text", "xml", "source", "rows", "floats", "gradients", "sparse zeros", "noise"]


def one_kind(t, n):
    """n bytes of member kind t: the members mixed_corpus would make for k = t, t + 8, ... (1 MiB each)."""
    out = bytearray()
    k = t
    while len(out) < n:
        # a mix of 12 MiB has members of 1 MiB: kinds 0 .. 6 at [t MiB, (t + 1) MiB), the noise (a quarter member) behind them
        piece = G.mixed_corpus(12 << 20, seed=G.SEED + 17 * k)
        out += piece[t << 20:(t + 1) << 20] if t < 7 else piece[7 << 20:(7 << 20) + (1 << 18)]
        k += 8
    return bytes(out[:n])


ctx = hip.Context(0)
for t, name in enumerate(NAMES):
    data = one_kind(t, N)
    d = hip.to_device(data)
    best = None
    for rep in range(2):
        got, info = ctx.debug_parse(d, N, hip.make_params(5, 22, 131072, 1 << 30))
        best = info if best is None or info["ms_parse"] < best["ms_parse"] else best
    print("KIND %-13s parse=%7.1f ms  index=%6.1f ms  searches=%d exact=%d commands=%d" % (
        name, best["ms_parse"], best["ms_index"], best["searches"], best.get("exact_searches", 0), best["commands"]), flush=True)
    del d
