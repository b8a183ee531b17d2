"""Offline fuzz of partition plans through the boundary on the simulator: random quality / window /
shard size / input / dictionaries; the reference driven with the same plan (one instance per shard, the
dictionaries attached to each) must give the same bytes.  python tools/fuzz_plan_sim.py SEED COUNT"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gen_inputs as G  # noqa: E402
from refharness import TABLES, Ref  # noqa: E402
from test_gpu_abi import _bind, drive  # noqa: E402
from test_sim_kernels import _fuzz_input  # noqa: E402

os.environ["BROTLI_AMD_TABLES"] = TABLES
sim = _bind(os.path.join(ROOT, "tests", "simt", "libbrotlienc_sim.so"))
ref = Ref()


def one(seed):
    rng = np.random.default_rng(seed)
    quality = int(rng.choice([2, 3, 4, 5, 5, 6, 7, 9]))
    lgwin = int(rng.choice([10, 14, 16, 17, 18, 20, 22, 24]))
    pieces = [_fuzz_input(rng) for _ in range(int(rng.integers(1, 4)))]
    pieces.append(G.enwik_text(int(rng.integers(20000, 120000)), seed=seed, vocab=3000))
    data = b"".join(bytes(p) for p in pieces)
    shard = int(rng.choice([4096, 20000, 1 << 15, 50000, 1 << 16, 100000, 1 << 17]))
    dictionaries = []
    if rng.integers(0, 2):
        for _ in range(int(rng.integers(1, 3))):
            a = int(rng.integers(0, max(1, len(data) - 32)))
            dictionaries.append(data[a:a + int(rng.integers(1, 30000))] if rng.integers(0, 2)
                                else bytes(G.enwik_text(int(rng.integers(1, 30000)), seed=seed + 3, vocab=3000)))
    want = ref.encode_plan(data, quality, lgwin, shard, dictionaries=dictionaries)
    params = ((1, quality), (2, lgwin), (5, min(len(data), 1 << 30)), (0x4D490001, shard))
    try:
        got, fin = drive(sim, data, [(len(data), 2)], params, dictionaries=dictionaries)
    except AssertionError:
        got, fin = None, False
    ok = fin and got == want
    print("seed %d q%d lgwin %d len %d shard %d dicts %s: %s" % (
        seed, quality, lgwin, len(data), shard, [len(d) for d in dictionaries], "ok" if ok else "MISMATCH"), flush=True)
    return ok


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    print("mismatching seeds:", [s for s in range(first, first + count) if not one(s)])
