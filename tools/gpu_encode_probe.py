"""GPU probe: full-pipeline parity vs the oracle + stage timing sweep."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gen_inputs as G
from brotli_amd import hip
from refharness import Oracle, Ref, have_ref

ctx = hip.Context(0)
o = Oracle()
ref = Ref() if have_ref() else None
alice = open(os.path.join(ROOT, "tests/golden/alice29.txt"), "rb").read()
text = G.enwik_text(4 << 20, seed=11, vocab=20000)
cases = [
    ("alice", alice, 0),
    ("tiny9", b"123456789", 0),
    ("x", b"x", 0),
    ("zeros", bytes(300000), 0),
    ("rle", (b"abcdefgh" * 50000)[:333333], 0),
    ("rand64k", G.random_bytes(1 << 16), 0),
    ("text4m/S", text, 0),
    ("text4m/256k", text, 1 << 18),
    ("text4m/100000", text, 100000),
    ("mixed2m/128k", G.mixed_corpus(2 << 20), 1 << 17),
    ("text_rand/64k", text[:200000] + G.random_bytes(150000) + text[:100000], 1 << 16),
]
if not os.environ.get("PROBE_SKIP_PARITY"):
    for name, data, shard in cases:
        want = o.encode_plan(data, 5, 22, shard)
        t0 = time.time()
        got, info = ctx.encode_host(data, hip.make_params(5, 22, shard))
        ok = got == want
        print("PARITY", name, len(data), len(want), len(got), ok, "%.2fs" % (time.time() - t0),
              {k: round(v, 2) if isinstance(v, float) else v for k, v in info.items()}, flush=True)
        if ok and ref is not None and len(data) < (8 << 20):
            assert ref.decompress(got, len(data)) == data
N = int(os.environ.get("PROBE_MB", "256")) << 20
t0 = time.time(); data = G.enwik_text(N); print("gen %.1fs" % (time.time() - t0), flush=True)
d_in = hip.to_device(data)
res = []
for shard in [int(x) for x in os.environ.get("PROBE_SHARDS", "1048576,262144,131072,65536").split(",")]:
    p = hip.make_params(5, 22, shard)
    cap = ctx.max_output(N, p)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
    for rep in range(2):
        n, info = ctx.encode_device(d_in, N, p, d_out)
    print("SWEEP shard=%d nshards=%d out=%d ratio=%.3f total=%.1fms init=%.1f parse=%.1f build=%.1f store=%.1f gather=%.1f -> %.0f MB/s" % (
        shard, info["nshards"], n, N / n, info["ms_total"], info["ms_init"], info["ms_parse"], info["ms_build"],
        info["ms_store"], info["ms_gather"], N / 1e6 / (info["ms_total"] / 1e3)), flush=True)
    res.append({"shard": shard, **info})
    if shard == 262144 and ref is not None and N <= (256 << 20):
        comp = d_out[:n].cpu().numpy().tobytes()
        t0 = time.time(); back = ref.decompress(comp, N); print("roundtrip ok=%s %.1fs" % (back == data, time.time() - t0), flush=True)
    del d_out
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "encode_probe.json"), "w"), indent=1)
