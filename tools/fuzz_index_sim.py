"""Offline fuzz of the indexed quality-5 parse (k_index.h + k_chain.h: the headline path) on the simulator:
random inputs / shard sizes / hashers / wave layouts / lane orders against the oracle's plan.
    python tools/fuzz_index_sim.py SEED COUNT"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gen_inputs as G  # noqa: E402
from refharness import Oracle  # noqa: E402
from simharness import Sim  # noqa: E402
from test_sim_kernels import IX_LAYOUTS, _fuzz_input, _oracle_plan  # noqa: E402

def make_case(seed):
    """(data, shard size, size hint, lane order) of a seed."""
    rng = np.random.default_rng(seed)
    pieces = [_fuzz_input(rng) for _ in range(int(rng.integers(1, 4)))]
    if rng.integers(0, 3) == 0:
        pieces.append(bytes(G.enwik_text(int(rng.integers(2000, 40000)), seed=seed, vocab=int(rng.choice([50, 3000])))))
    if rng.integers(0, 4) == 0:
        pieces.append(bytes(int(rng.integers(1, 30000))))            # a run of zeros: one key run
    data = b"".join(pieces)
    shard = int(rng.choice([0, 0, 300, 1000, 2500, 7000, 20000]))
    hint = (1 << 30) if rng.integers(0, 2) else 0                     # H68 vs H58
    rev = int(rng.integers(0, 2))
    return data, shard, hint, rev


def one(seed, sim, oracle):
    data, shard, hint, rev = make_case(seed)
    want = _oracle_plan(oracle, data, hint, shard)
    bad = []
    for layout, flags in IX_LAYOUTS.items():
        for extra in (0, 4):
            if sim.encode(data, size_hint=hint, shard_size=shard, reverse=rev, flags=flags | extra) != want:
                bad.append((layout, extra))
    print("seed %d len %d shard %d hint %d rev %d: %s" % (seed, len(data), shard, hint, rev, "ok" if not bad else ("MISMATCH", bad)), flush=True)
    return not bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sim, oracle = Sim(), Oracle()
    print("mismatching seeds:", [s for s in range(first, first + count) if not one(s, sim, oracle)])
