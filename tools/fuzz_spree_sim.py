"""Offline fuzz of the literal spree under the indexed parse and of the tainted-result rule (IX_FULLRUN, k_chain.h) on the
simulator: noise / small alphabets / floats with echoes, random shard sizes, both hashers, every wave layout, lanes in
either order, against the oracle's plan.
    python tools/fuzz_spree_sim.py SEED COUNT"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from refharness import Oracle  # noqa: E402
from simharness import Sim  # noqa: E402
from test_sim_kernels import IX_LAYOUTS, _noise_with_echoes, _oracle_plan  # noqa: E402


def make_case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20000, 160000))
    echo = int(rng.choice([60, 200, 900, 4000, 20000]))
    vocab = int(rng.choice([0, 0, 0, 3, 7, 40, 200]))
    data = _noise_with_echoes(n, seed, echo, vocab)
    if rng.integers(0, 3) == 0:       # a stretch of floats (repeating high bytes: short matches, tainted successors)
        f = np.cumsum(rng.normal(size=int(rng.integers(500, 8000)))).astype(np.float32).tobytes()
        k = int(rng.integers(0, len(data)))
        data = data[:k] + f + data[k:]
    shard = int(rng.choice([0, 0, 9000, 30000, 66000, 70000]))
    hint = (1 << 30) if rng.integers(0, 2) else 0
    rev = int(rng.integers(0, 2))
    return data, shard, hint, rev


def one(seed, sim, oracle):
    data, shard, hint, rev = make_case(seed)
    want = _oracle_plan(oracle, data, hint, shard)
    bad = [layout for layout, flags in IX_LAYOUTS.items()
           if sim.encode(data, size_hint=hint, shard_size=shard, reverse=rev, flags=flags) != want]
    print("seed %d len %d shard %d hint %d rev %d: %s" % (seed, len(data), shard, hint, rev, "ok" if not bad else ("MISMATCH", bad)), flush=True)
    return not bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sim, oracle = Sim(), Oracle()
    print("mismatching seeds:", [s for s in range(first, first + count) if not one(s, sim, oracle)])
