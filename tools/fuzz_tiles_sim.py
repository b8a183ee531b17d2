"""Fuzz of the TILED quality-5 chain (k_chain.h tiles / sweeps, k_tile.h) on the simulator: inputs of a few tiles
built from pieces that make the tiles depend on each other — text over small vocabularies (full bucket windows),
repeats across tile boundaries, runs (clipped store ranges), incompressible stretches (the literal spree), real
English (the static-dictionary gate stays open: the shard must leave the tiled path) — as one shard or a plan,
lanes and workgroups in either order, against the oracle's plan.
    python tools/fuzz_tiles_sim.py SEED COUNT"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gen_inputs as G  # noqa: E402
from refharness import Oracle  # noqa: E402
from simharness import Sim  # noqa: E402
from test_sim_kernels import _fuzz_input, _oracle_plan  # noqa: E402

ALICE = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()
TILE = 1 << 16


def make_case(seed):
    rng = np.random.default_rng(1000003 * seed + 17)
    target = int(rng.integers(TILE + 2000, 4 * TILE + 30000))
    out = bytearray()
    flavour = int(rng.integers(0, 5))
    while len(out) < target:
        k = int(rng.integers(0, 10))
        if k <= 3 or flavour == 0:
            out += G.enwik_text(int(rng.integers(3000, 90000)), seed=int(rng.integers(0, 1 << 30)),
                                vocab=int(rng.choice([40, 300, 3000, 50000])))
        elif k == 4:
            out += rng.integers(0, 256, int(rng.integers(50, 5000 if flavour != 1 else 40000)), dtype=np.uint8).tobytes()
        elif k == 5 and len(out) > 100:
            d = int(rng.integers(1, min(len(out), 200000)))          # a long repeat, often across a tile boundary
            n = int(rng.integers(10, 30000))
            for _ in range(n):
                out.append(out[-d])
        elif k == 6:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(10, 3000))
        elif k == 7 and flavour == 2:
            a = int(rng.integers(0, len(ALICE) - 20000))
            out += ALICE[a:a + int(rng.integers(2000, 20000))]
        elif k == 8:
            out += _fuzz_input(rng)
        else:
            out += G.mixed_corpus(int(rng.integers(2000, 30000)), seed=int(rng.integers(0, 1 << 20)))
    data = bytes(out[:target])
    if flavour == 3:                                                  # land a boundary inside a run / a repeat
        b = bytearray(data)
        for t in range(1, len(b) // TILE + 1):
            p = t * TILE + int(rng.integers(-40, 40))
            b[max(0, p - 300):p + 300] = bytes([b[max(0, p - 301)]]) * len(b[max(0, p - 300):p + 300])
        data = bytes(b[:target])
    shard = int(rng.choice([0, 0, TILE + 1234, 2 * TILE + 7, 150000]))
    rev = int(rng.integers(0, 2))
    warm = int(rng.choice([256, 1024, 2048]))
    return data, shard, rev, warm


def case_hint(seed, data):
    """H68 (a MiB announced) for two seeds of three, else the hint the library derives itself: below a MiB that is H58,
    the 4-byte hasher with its own block-tail rules (hash_longest_match_simd_inc.h)."""
    return (1 << 30) if seed % 3 else 0


def one(seed, sim, oracle, verbose=True):
    data, shard, rev, warm = make_case(seed)
    hint = case_hint(seed, data)
    want = _oracle_plan(oracle, data, hint, shard)
    os.environ["SIM_TILE_KB"] = "64"
    os.environ["SIM_TILE_WARM"] = str(warm)
    os.environ["SIM_SWEEP_GROUPS"] = str((1, 2, 4)[seed % 3 if seed % 2 else 1])     # (2 = the library's default)
    try:
        got = sim.encode(data, 5, 22, hint, shard, reverse=rev, flags=2 | 64)
    finally:
        del os.environ["SIM_TILE_KB"], os.environ["SIM_TILE_WARM"], os.environ["SIM_SWEEP_GROUPS"]
    ok = got == want
    if verbose:
        print("seed %d len %d shard %d rev %d warm %d hint %d: %s" % (seed, len(data), shard, rev, warm, hint, "ok" if ok else "MISMATCH"), flush=True)
    return ok


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sim, oracle = Sim(), Oracle()
    print("mismatching seeds:", [s for s in range(first, first + count) if not one(s, sim, oracle)])
