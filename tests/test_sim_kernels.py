"""Kernel LOGIC on the host SIMT simulator (tests/simt) against the oracle: the
same headers that hipcc compiles for gfx950 are compiled for the host with the
wave primitives replaced by a 64-fiber rendezvous scheduler.  No GPU needed;
sizes are small because the simulator switches fibers at every cross-lane op.
The `-m gpu` tests run the real kernels at full sizes."""
import numpy as np
import pytest

import gen_inputs as G

import os  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
from simharness import Sim, oracle_commands

ALICE = open(__file__.replace("test_sim_kernels.py", "golden/alice29.txt"), "rb").read()


@pytest.fixture(scope="module")
def sim():
    return Sim()


def _oracle_plan(oracle, data, hint, shard):
    n = len(data)
    shard = shard or n
    parts, off = [], 0
    while off < n:
        m = min(shard, n - off)
        parts.append(oracle.encode_shard(data[off:off + m], 5, 22, hint or min(n, 1 << 30),
                                         min(off, 1 << 30), off + m == n))
        off += m
    return b"".join(parts)


def test_parse_commands_alice_prefix(sim, oracle):
    data = ALICE[:60000]
    want = oracle_commands(oracle, data, 5, 22, 0, 0)
    got, stats = sim.parse(data, 5, 22, 0, 0)
    assert np.array_equal(want, got)
    assert stats[0] > 0 and stats[1] > 0


@pytest.mark.parametrize("reverse,no_pair", [(1, 0), (0, 1)])
def test_parse_lane_order_and_unpaired(sim, oracle, reverse, no_pair):
    """Lanes scheduled high-to-low (catches missing wave_sync) and the search
    without the speculative (p, p+1) pair must give the same commands."""
    data = G.enwik_text(40000, seed=7, vocab=5000)
    want = oracle_commands(oracle, data, 5, 22, 1 << 30, 1 << 15)
    got, _ = sim.parse(data, 5, 22, 1 << 30, 1 << 15, reverse=reverse, no_pair=no_pair)
    assert np.array_equal(want, got)


CASES = {
    "alice_48k": (ALICE[:48000], 0, 0),                                  # H58, 1 context
    "text_hint_2shards": (G.enwik_text(1 << 16, seed=11, vocab=20000), 1 << 30, 1 << 15),  # H68, 13 contexts, flint
    "ragged_shards": (G.enwik_text(50000, seed=5, vocab=3000), 0, 17000),
    "mixed": (G.mixed_corpus(1 << 16)[:50000], 0, 0),
    "random_raw": (G.random_bytes(20000), 0, 0),
    "text_then_random": (G.enwik_text(20000, seed=2) + G.random_bytes(20000), 0, 0),
    "zeros": (bytes(70000), 0, 0),
    "rle": ((b"abcdefgh" * 9000)[:70001], 0, 0),
    "tiny1": (b"x", 0, 0), "tiny2": (b"xy", 0, 0), "tiny3": (b"xyz", 0, 0),
    "tiny9": (b"123456789", 0, 0), "x64": (b"x" * 64, 0, 0),
    "shards_of_1_2_3": (b"abcabcabcabc", 0, 3),
    # runs of one byte between random stretches: up to 32 insertions of one bucket key in a step
    "runs": (b"".join(G.random_bytes(37 + 11 * i, seed=i) + bytes([65 + i]) * (20 + 7 * i) for i in range(40)), 0, 0),
}


@pytest.mark.parametrize("name", list(CASES))
def test_encode_bytes_match_oracle(sim, oracle, name):
    data, hint, shard = CASES[name]
    assert sim.encode(data, 5, 22, hint, shard) == _oracle_plan(oracle, data, hint, shard)


@pytest.mark.parametrize("flags", [2, 6])
@pytest.mark.parametrize("name", ["text_hint_2shards", "ragged_shards", "text_then_random",
                                  "zeros", "rle", "tiny3", "shards_of_1_2_3"])
def test_quad_kernel_bytes_match_oracle(sim, oracle, name, flags):
    """Four shards per wave (k_parse4.h): arg-max resolve (flags=2) and the
    step-by-step resolve (flags=6) both reproduce the oracle."""
    data, hint, shard = CASES[name]
    assert sim.encode(data, 5, 22, hint, shard, flags=flags) == _oracle_plan(oracle, data, hint, shard)


def _oracle_plan_q(oracle, data, quality, lgwin, hint, shard):
    n = len(data)
    shard = shard or n
    parts, off = [], 0
    while off < n:
        m = min(shard, n - off)
        parts.append(oracle.encode_shard(data[off:off + m], quality, lgwin, hint or min(n, 1 << 30),
                                         min(off, 1 << 30), off + m == n))
        off += m
    return b"".join(parts)


@pytest.mark.parametrize("quality,lgwin", [(6, 22), (7, 22), (8, 20), (9, 24)])
@pytest.mark.parametrize("name", ["text_hint_2shards", "mixed", "text_then_random", "shards_of_1_2_3"])
def test_deep_kernel_bytes_match_oracle(sim, oracle, name, quality, lgwin):
    """k_parse_deep.h: 32-slot tagged buckets (q6) and the H5 / H6 hashers with
    64 / 128 / 256 slots and 10 / 16 distance-cache probes (q7 - q9)."""
    data, hint, shard = CASES[name]
    assert sim.encode(data, quality, lgwin, hint, shard) == _oracle_plan_q(oracle, data, quality, lgwin, hint, shard)


@pytest.mark.parametrize("quality", [6, 9])
def test_deep_kernel_step_by_step_resolve(sim, oracle, quality):
    data = G.enwik_text(30000, seed=31, vocab=2000)
    assert sim.encode(data, quality, 22, 1 << 30, 0, flags=4, reverse=1) == \
        _oracle_plan_q(oracle, data, quality, 22, 1 << 30, 0)


@pytest.mark.parametrize("groups", [1, 2])
def test_quad_kernel_fewer_shards_per_wave(sim, oracle, groups):
    """JOB_FLAG_GROUPS: one or two 16-lane groups of a wave own a shard, the others idle."""
    data, shard, hint = CASES["ragged_shards"]
    assert sim.encode(data, size_hint=hint, shard_size=shard, flags=2 | (groups << 8)) == \
        oracle.encode_plan(data, 5, 22, shard) if not hint else True
    data = G.enwik_text(120000, seed=77, vocab=5000)
    assert sim.encode(data, shard_size=30000, flags=2 | (groups << 8)) == oracle.encode_plan(data, 5, 22, 30000)


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("name", ["text_hint_2shards", "ragged_shards", "mixed", "text_then_random",
                                  "shards_of_1_2_3", "alice_48k", "zeros", "rle", "x64", "random_raw", "runs"])
def test_quad_kernel_scout_groups(sim, oracle, name, groups):
    """JOB_FLAG_DUO: a second group per shard searches the following position in the same
    step; its result is used only when the state machine asks for exactly that position and
    the two positions hash to different buckets."""
    data, shard, hint = CASES[name]
    want = oracle.encode_plan(data, 5, 22, shard) if not hint else None
    for reverse in (0, 1):
        got = sim.encode(data, size_hint=hint, shard_size=shard, reverse=reverse, flags=2 | 32 | (groups << 8))
        if want is None:
            want = sim.encode(data, size_hint=hint, shard_size=shard, flags=2)
        assert got == want


def test_quad_kernel_scout_groups_forced_slow_resolve(sim, oracle):
    data = G.enwik_text(200000, seed=78, vocab=3000)
    assert sim.encode(data, shard_size=50000, flags=2 | 4 | 32 | (2 << 8)) == oracle.encode_plan(data, 5, 22, 50000)


# ---- indexed quality-5 parse (k_index.h + k_chain.h): JOB_FLAG_INDEXED = 64 ----
IX_LAYOUTS = {
    "groups4": 2 | 64,                      # four shards per wave, 16 lanes each (the product default for many shards)
    "groups2": 2 | 64 | (2 << 8),
    "groups1": 2 | 64 | (1 << 8),
}


@pytest.mark.parametrize("layout", list(IX_LAYOUTS))
@pytest.mark.parametrize("name", list(CASES))
def test_indexed_parse_bytes_match_oracle(sim, oracle, name, layout):
    """The position index + table-free chain reproduce the oracle on every case of the table
    kernels, in every wave layout, with lanes scheduled in either order."""
    data, hint, shard = CASES[name]
    want = _oracle_plan(oracle, data, hint, shard)
    for reverse in ((0, 1) if layout == "groups4" else (0,)):
        assert sim.encode(data, 5, 22, hint, shard, reverse=reverse, flags=IX_LAYOUTS[layout]) == want


@pytest.mark.parametrize("name,layout", [("alice_48k", "groups4"), ("text_hint_2shards", "groups4"),
                                         ("text_hint_2shards", "groups1"), ("mixed", "groups2"), ("rle", "groups4"),
                                         ("runs", "groups1"), ("shards_of_1_2_3", "groups4")])
def test_indexed_parse_forced_exact_search(sim, oracle, name, layout):
    """JOB_FLAG_FORCE_SLOW: every search goes through c_search_exact (the sorted array + the
    bitmap of unstored positions) and the step-by-step resolve."""
    data, hint, shard = CASES[name]
    assert sim.encode(data, 5, 22, hint, shard, flags=IX_LAYOUTS[layout] | 4) == _oracle_plan(oracle, data, hint, shard)


def test_indexed_parse_english_text_dictionary_gate_open(sim, oracle):
    """alice29.txt keeps the static-dictionary gate open (hash.h:179-202): misses and empty
    lazy probes must leave the group fast path."""
    want = oracle.encode_plan(ALICE, 5, 22, 40000)
    for layout in ("groups4", "groups2", "groups1"):
        assert sim.encode(ALICE, 5, 22, 0, 40000, flags=IX_LAYOUTS[layout]) == want, layout


def test_indexed_parse_long_shards_h68(sim, oracle):
    """Two 160 KiB shards with the 5-byte hasher: buckets fill up (16 slots), long copies."""
    data = G.enwik_text(320 << 10, seed=21, vocab=4000) 
    want = _oracle_plan(oracle, data, 1 << 30, 160 << 10)
    for layout in ("groups4", "groups1"):
        assert sim.encode(data, 5, 22, 1 << 30, 160 << 10, flags=IX_LAYOUTS[layout]) == want, layout


def test_indexed_parse_copy_ends_at_block_end(sim, oracle):
    """Regression: a copy that ends exactly at the end of an input block; the next block's
    stitch (..64_simd_inc.h:139-151) still stores the last three positions of the block."""
    data = G.enwik_text(4 << 20, seed=11, vocab=20000)[2 << 18:3 << 18]
    want = oracle.encode_shard(data, 5, 22, 4 << 20, 2 << 18, False)
    assert sim.encode(data, 5, 22, 4 << 20, 0, stream_base=2 << 18, is_last=False, flags=IX_LAYOUTS["groups1"]) == want


def test_indexed_parse_key_run_past_the_16_bit_store_counter(sim, oracle):
    """Sparse zeros: one key run of > 65520 positions, so the wrap of the reference's 16-bit
    store counter (..64_simd_inc.h:250-257) is in reach and every search there is an exact one;
    the number of stores of the run is carried from search to search (c_search_exact) — without
    that this shard takes minutes here and seconds on the GPU.  The second input stores more
    than 65536 positions of the run: the counter does wrap."""
    import time
    rng = np.random.default_rng(7)
    z = np.zeros(140000, dtype=np.uint8)
    pos = rng.integers(0, z.size, size=z.size // 50)
    z[pos] = rng.integers(1, 256, size=pos.size)
    t0 = time.time()
    assert sim.encode(z.tobytes(), 5, 22, 1 << 30, 0, flags=IX_LAYOUTS["groups4"]) == _oracle_plan(oracle, z.tobytes(), 1 << 30, 0)
    assert time.time() - t0 < 60
    z = np.zeros(260000, dtype=np.uint8)
    pos = rng.integers(0, z.size, size=z.size // 6)
    z[pos] = rng.integers(1, 4, size=pos.size)
    assert sim.encode(z.tobytes(), 5, 22, 0, 0, flags=IX_LAYOUTS["groups1"]) == _oracle_plan(oracle, z.tobytes(), 0, 0)


def _spree_members():
    """Members of the Silesia-style mix on which the literal spree leaves most positions unstored (floats, noise)
    and the ones around them, 96 KiB each."""
    big = G.mixed_corpus(8 << 20)
    member = (8 << 20) // 12 + 1
    return {name: big[k * member + 1000:k * member + 1000 + (96 << 10)]
            for k, name in ((4, "floats"), (5, "gradient"), (6, "zeros"), (7, "noise"))}


@pytest.mark.parametrize("name", ["floats", "gradient", "zeros", "noise"])
def test_indexed_parse_tainted_results_that_still_hold(sim, oracle, name):
    """IX_FULLRUN (k_index_layout.h): a search whose key run has <= 16 predecessors keeps its index result although
    some of them were not stored — when it found nothing, or when its winner is a stored position (c_taint_holds).
    Floats and noise: the literal spree taints nearly every search.  Both hashers, two wave layouts; the forced
    exact search (which never uses the rule) gives the same bytes."""
    data = _spree_members()[name]
    for hint in (1 << 30, 0):
        want = _oracle_plan(oracle, data, hint, 48 << 10)
        for layout in ("groups4", "groups1"):
            assert sim.encode(data, 5, 22, hint, 48 << 10, flags=IX_LAYOUTS[layout]) == want, (hint, layout)
    assert sim.encode(data, 5, 22, 1 << 30, 48 << 10, flags=IX_LAYOUTS["groups4"] | 4) == _oracle_plan(oracle, data, 1 << 30, 48 << 10)


def _noise_with_echoes(n, seed, echo_every=900, vocab=0):
    """Noise in which earlier pieces come back (5 .. 60 bytes, from near and far): the literal spree runs, is
    interrupted by matches — distance-cache ones too: the same piece twice in a row — and starts again."""
    rng = np.random.default_rng(seed)
    out = bytearray(rng.integers(0, 256, size=3000, dtype=np.uint8).tobytes())
    while len(out) < n:
        out += rng.integers(0, 256 if not vocab else vocab, size=int(rng.integers(20, 2 * echo_every)), dtype=np.uint8).tobytes()
        ln = int(rng.integers(5, 60))
        src = int(rng.integers(0, len(out) - ln)) if rng.integers(0, 3) else max(0, len(out) - int(rng.integers(ln, 400)))
        piece = bytes(out[src:src + ln])
        out += piece
        if rng.integers(0, 4) == 0:
            out += rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8).tobytes() + piece
    return bytes(out[:n])


@pytest.mark.parametrize("seed,n,shard,echo,vocab", [(1, 150000, 0, 900, 0), (2, 200000, 70000, 300, 0), (4, 100000, 30000, 200, 7)])
def test_indexed_parse_literal_spree_steps(sim, oracle, seed, n, shard, echo, vocab):
    """The literal spree (backward_references_inc.h:208-236) under the indexed parse: noise with echoes — most
    positions unstored, nearly every search tainted, matches that interrupt the spree —, block ends inside it (64 KiB
    blocks, shards of one and of several blocks), both hashers, small alphabets (key runs longer than 16: no
    IX_FULLRUN), lanes in either order; equal to the forced exact search.  (Written for a variant of the chain's fast
    path that took 16 spree searches per step — measured on the MI355X: +3 ms per GiB of text for nothing on the
    mix, whose chain time is its sparse-zero shards', profiles/r04_e — and kept for what it covers.)"""
    data = _noise_with_echoes(n, seed, echo, vocab)
    for hint in (1 << 30, 0):
        want = _oracle_plan(oracle, data, hint, shard)
        for layout, rev in (("groups4", 0), ("groups4", 1), ("groups1", 0)):
            assert sim.encode(data, 5, 22, hint, shard, reverse=rev, flags=IX_LAYOUTS[layout]) == want, (hint, layout, rev)
    assert sim.encode(data, 5, 22, 1 << 30, shard, flags=IX_LAYOUTS["groups2"] | 4) == _oracle_plan(oracle, data, 1 << 30, shard)


@pytest.mark.parametrize("seed", range(3))
def test_indexed_parse_fuzz(sim, oracle, seed):
    rng = np.random.default_rng(4000 + seed)
    for _ in range(6):
        data = _fuzz_input(rng)
        shard = int(rng.integers(0, 3)) * int(rng.integers(300, 2500))
        hint = (1 << 30) if rng.integers(0, 2) else 0          # H68 vs H58
        want = _oracle_plan(oracle, data, hint, shard)
        rev = int(rng.integers(0, 2))
        for layout in IX_LAYOUTS:
            assert sim.encode(data, size_hint=hint, shard_size=shard, reverse=rev, flags=IX_LAYOUTS[layout]) == want, \
                (seed, len(data), shard, hint, layout)
        assert sim.encode(data, size_hint=hint, shard_size=shard, reverse=rev, flags=IX_LAYOUTS["groups4"] | 4) == want


@pytest.mark.parametrize("seed", [431, 1011, 1638])
def test_indexed_parse_dictionary_gate_at_the_spree_boundary(sim, oracle, seed):
    """Cases of tools/fuzz_index_sim.py that differed: no match at the first position of a fast step, the static
    dictionary still being consulted, and the literal spree tripping right behind it — the step goes to the generic
    path, which searches the position itself; the fast path must not have asked the dictionary already (two lookups
    counted twice close the gate earlier than the reference does, and a later dictionary match is missed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_index_sim", os.path.join(os.path.dirname(HERE), "tools", "fuzz_index_sim.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    data, shard, hint, rev = fz.make_case(seed)
    if seed == 431:
        data = data[:600]
    want = _oracle_plan(oracle, data, hint, shard)
    for layout in IX_LAYOUTS:
        assert sim.encode(data, size_hint=hint, shard_size=shard, reverse=rev, flags=IX_LAYOUTS[layout]) == want, layout


def test_quad_kernel_many_shards_reverse(sim, oracle):
    """7 shards over 2 waves (one group idle), lanes scheduled high-to-low."""
    data = G.enwik_text(70000, seed=13, vocab=3000)
    assert sim.encode(data, 5, 22, 1 << 30, 10000, reverse=1, flags=2) == \
        _oracle_plan(oracle, data, 1 << 30, 10000)


def test_encode_reverse_lane_order(sim, oracle):
    data = G.enwik_text(30000, seed=9, vocab=4000)
    assert sim.encode(data, 5, 22, 1 << 30, 0, reverse=1) == _oracle_plan(oracle, data, 1 << 30, 0)


def test_rank_piece_is_slice_of_whole_plan(sim, oracle):
    """Multi-GPU contract: a rank that encodes the middle piece of a stream
    (stream_base > 0, is_last = 0) produces exactly that slice of the whole
    plan's output."""
    data = G.enwik_text(60000, seed=4, vocab=4000)
    shard, hint = 10000, 60000
    whole = _oracle_plan(oracle, data, hint, shard)
    pieces = [sim.encode(data[a:b], 5, 22, hint, shard, stream_base=a, is_last=(b == len(data)))
              for a, b in ((0, 20000), (20000, 40000), (40000, 60000))]
    assert b"".join(pieces) == whole


# --- quality 1: k_fast_* (two-pass fragment compressor) -------------------------------------------

def _fast_inputs():
    text = G.enwik_text(400000, seed=5, vocab=20000)
    yield "alice", ALICE
    yield "random", G.random_bytes(400000, seed=1)                      # raw blocks, fragment rewritten raw
    yield "text_rand", text[:200000] + G.random_bytes(150000, seed=2) + text[:100000]   # raw block inside a compressed fragment
    yield "mixed", G.mixed_corpus(1 << 19)
    yield "zeros", bytes(300000)                                         # one literal symbol, long copies
    yield "rle", (b"abcdefgh" * 50000)[:333333]
    for n in (1, 2, 15, 16, 17, 257, 32768, 32769, 131072, 131073):
        yield "text%d" % n, G.enwik_text(n, seed=n, vocab=2000)


FAST = dict(_fast_inputs())


@pytest.mark.parametrize("name", list(FAST))
def test_fast_kernels_one_call(sim, oracle, name):
    """Whole input in one FINISH call: fragments of 1 << lgwin (table size and 4- or
    6-byte matches follow the fragment size), both lane orders."""
    data = FAST[name]
    for lgwin, reverse in ((22, 0), (16, 1), (10, 0)):
        if lgwin == 10 and len(data) > 200000:
            continue
        assert sim.encode_fast(data, lgwin, None, reverse) == oracle.encode_fast(data, lgwin), (lgwin, reverse)


@pytest.mark.parametrize("name", ["alice", "text_rand", "mixed", "text131073"])
@pytest.mark.parametrize("chunk", [65536, 100000])
def test_fast_kernels_call_sequences(sim, oracle, name, chunk):
    """CLI-style feeding: every call is its own run of fragments; the last call is an
    empty FINISH (a zero-length fragment that only carries ISLAST)."""
    data = FAST[name]
    calls, off = [], 0
    while off < len(data):
        m = min(chunk, len(data) - off)
        off += m
        calls.append((m, 0))
    calls.append((0, 2))
    assert sim.encode_fast(data, 22, calls) == oracle.encode_fast(data, 22, calls)


def test_fast_kernels_table_slots_are_reused(sim, oracle, monkeypatch):
    """More fragments than table slots: a wave re-zeroes its table between fragments."""
    monkeypatch.setenv("SIM_FAST_SLOTS", "2")
    data = FAST["mixed"]
    assert sim.encode_fast(data, 16) == oracle.encode_fast(data, 16)


# --- seeded fuzz: many small structured inputs through every kernel family ---------------------------

def _fuzz_input(rng):
    """Small inputs with the features that steer the encoder: repeats at assorted distances,
    runs, dictionary words, incompressible stretches."""
    words = [b"the ", b"and ", b"that ", b"with ", b"http://", b"</div>", b"function ", b"0000", b"\n\n", b"Time "]
    out = bytearray()
    target = int(rng.integers(1, 6000))
    while len(out) < target:
        k = int(rng.integers(0, 6))
        if k == 0:
            out += rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
        elif k == 1 and out:
            d = int(rng.integers(1, len(out) + 1))
            n = int(rng.integers(2, 70))
            for _ in range(n):
                out.append(out[-d])
        elif k == 2:
            out += bytes([int(rng.integers(32, 127))]) * int(rng.integers(1, 50))
        elif k == 3:
            out += words[int(rng.integers(0, len(words)))]
        elif k == 4:
            out += bytes(rng.integers(97, 123, int(rng.integers(1, 12)), dtype=np.uint8).tolist()) + b" "
        else:
            out += G.enwik_text(int(rng.integers(8, 200)), seed=int(rng.integers(0, 1 << 30)), vocab=300)
    return bytes(out[:target])


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_small_inputs_all_kernels(sim, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(9):
        data = _fuzz_input(rng)
        shard = int(rng.integers(0, 3)) * int(rng.integers(300, 2500))
        hint = (1 << 30) if rng.integers(0, 2) else 0          # H68 vs H58
        want = sim.encode(data, size_hint=hint, shard_size=shard, flags=2) if hint else oracle.encode_plan(data, 5, 22, shard)
        if hint:   # the oracle's plan driver derives the hint itself; pin the base kernel on its shard API
            parts, off = [], 0
            sh = shard or len(data)
            while off < len(data):
                m = min(sh, len(data) - off)
                parts.append(oracle.encode_shard(data[off:off + m], 5, 22, hint, off, off + m == len(data)))
                off += m
            assert want == b"".join(parts)
        rev = int(rng.integers(0, 2))
        for flags in (2 | 32 | (1 << 8), 2 | 32 | (2 << 8), 2 | 4 | 32 | (2 << 8)):
            assert sim.encode(data, size_hint=hint, shard_size=shard, reverse=rev, flags=flags) == want, (seed, len(data), shard, flags)
        q = int(rng.integers(6, 10))
        assert sim.encode(data, quality=q, lgwin=22, shard_size=shard) == oracle.encode_plan(data, q, 22, shard), (seed, q)
        lgwin = int(rng.choice([10, 12, 16, 18, 22]))
        assert sim.encode_fast(data, lgwin, None, rev) == oracle.encode_fast(data, lgwin), (seed, lgwin)


# --- one encoder instance fed call by call (the HIP layer's device-resident stream) -----------------

def _stream_chunks(n, size, flush_every=0):
    ops, off, i = [], 0, 0
    while off < n:
        m = min(size, n - off)
        off += m
        i += 1
        ops.append((m, 2 if off == n else (1 if flush_every and i % flush_every == 0 else 0)))
    return ops


# ---- qualities 2 - 4 (k_parse_quick.h; single-block / count-only tree modes of k_build / k_store) ----
def _plan_q(oracle, data, quality, lgwin, hint, shard):
    n = len(data)
    shard = shard or n
    parts, off = [], 0
    while off < n:
        m = min(shard, n - off)
        parts.append(oracle.encode_shard(data[off:off + m], quality, lgwin, hint or n, min(off, 1 << 30), off + m == n))
        off += m
    return b"".join(parts)


QUICK_TEXT = G.enwik_text(140000, seed=4, vocab=20000)
QUICK_CASES = {
    "text": QUICK_TEXT,
    "mixed": G.mixed_corpus(120000),
    "random": G.random_bytes(40000, seed=1),
    "zeros": bytes(100000),
    "rle": (b"abcdefgh" * 20000)[:111111],
    "text_rand": QUICK_TEXT[:50000] + G.random_bytes(30000, seed=2) + QUICK_TEXT[:40000],
    "x64": b"x" * 64,
    "t3": b"xyz",
    "t9": b"123456789",
    "hello": b"hello hello hello hello",       # quality 2 with <= 128 commands: the static command / distance codes
}


@pytest.mark.parametrize("name", list(QUICK_CASES))
@pytest.mark.parametrize("quality", [2, 3, 4])
def test_quick_hashers_bytes_match_oracle(sim, oracle, name, quality):
    """H2 / H3 / H4 (H54 once a MiB is announced), the lazy probe seeded with the length to beat,
    16 KiB input blocks and the early cut at qualities 2 - 3, one prefix code per category
    (count-only trees or the static codes at quality 2): one stream, shards with STREAM_OFFSET,
    windows below the shard length (10, 16 bits) and the largest one."""
    data = QUICK_CASES[name]
    for lgwin, hint, shard in ((22, 0, 0), (18, 1 << 30, 50000), (16, 0, 0), (10, 0, 0), (24, 1 << 30, 0)):
        if shard and len(data) < 1000:
            continue
        want = _plan_q(oracle, data, quality, lgwin, hint, shard)
        assert sim.encode(data, quality, lgwin, hint or len(data), shard) == want, (lgwin, hint, shard)


@pytest.mark.parametrize("quality,lgwin", [(2, 22), (3, 18), (4, 22), (4, 16)])
def test_quick_stream_call_sequences_equal_reference(sim, ref, quality, lgwin):
    """One encoder instance at qualities 2 - 4 resumed call after call (PROCESS, FLUSH, FINISH)."""
    text = G.enwik_text(120000, seed=54, vocab=20000)
    for hint in (1 << 20, 0):
        for calls in (_stream_chunks(len(text), 30000), _stream_chunks(len(text), 17000, 3)):
            want = ref.encode_calls(text, quality, lgwin, calls, size_hint=hint)
            assert sim.stream(text, calls, quality, lgwin, size_hint=hint) == want, (quality, hint, calls[:3])


FC_TEXT = G.enwik_text(200000, seed=5, vocab=20000)
FC_CASES = {
    "text_laps_ring": FC_TEXT,                          # 200 KB through a 128 KiB ring
    "mixed": G.mixed_corpus(150000),
    "text_rand": FC_TEXT[:60000] + G.random_bytes(30000, seed=2) + FC_TEXT[:40000],
    "rle": (b"abcdefgh" * 20000)[:111111],
    "zeros": bytes(100000),
    "t9": b"123456789",
}


@pytest.mark.parametrize("name", list(FC_CASES))
@pytest.mark.parametrize("quality,lgwin,shard", [(5, 16, 0), (6, 10, 0), (7, 14, 70000), (8, 12, 0), (9, 16, 0), (9, 10, 50000)])
def test_forgetful_chain_hashers_bytes_match_oracle(sim, oracle, name, quality, lgwin, shard):
    """Qualities 5 - 9 at windows of 10 - 16 bits: H40 / H41 / H42 in k_parse_quick.h (chains
    through banks of recycled slots, tiny-hash filter on 4 / 10 / 16 distance-cache entries,
    ring-end rules of the chain walk), one stream and shards."""
    data = FC_CASES[name]
    if shard and len(data) < 1000:
        pytest.skip("one shard")
    want = _plan_q(oracle, data, quality, lgwin, 1 << 30 if shard else 0, shard)
    assert sim.encode(data, quality, lgwin, (1 << 30) if shard else len(data), shard) == want


@pytest.mark.parametrize("quality,lgwin", [(5, 16), (9, 12)])
def test_forgetful_chain_stream_call_sequences_equal_reference(sim, ref, quality, lgwin):
    text = G.enwik_text(150000, seed=55, vocab=20000)
    for calls in (_stream_chunks(len(text), 40000), _stream_chunks(len(text), 23000, 2)):
        want = ref.encode_calls(text, quality, lgwin, calls, size_hint=0)
        assert sim.stream(text, calls, quality, lgwin, size_hint=0) == want, (quality, calls[:3])


@pytest.mark.parametrize("quality,lgwin", [(9, 24)])      # (5, 22) and (6, 22): tests/test_abi_on_sim.py
def test_stream_call_sequences_equal_reference(sim, ref, quality, lgwin):
    """k_parse (quality 5) / k_parse_deep (6-9) resumed call after call with the state the
    previous call left: PROCESS feeds, FLUSHes, FINISH; metadata blocks continue the open byte
    (final_op 3: flush without the padding block).  Checked against the reference library."""
    text = G.enwik_text(150000, seed=53, vocab=20000)
    meta = b"\x01\x02\x03" * 50
    hint = 1 << 20
    for data, calls in (
            (text, _stream_chunks(len(text), 40000, 2)),
            (text[:60000] + meta + text[60000:150000], [(60000, 0), (len(meta), 3), (90000, 2)]),
            (text[:90000], [(30000, 1), (0, 3), (60000, 2)])):
        want = ref.encode_calls(data, quality, lgwin, calls, size_hint=hint)
        assert sim.stream(data, calls, quality, lgwin, size_hint=hint) == want, (quality, calls)


def test_stream_unknown_operation_fails_instead_of_spinning(sim):
    """A final_op the driver does not know must end in an error, not in an endless block loop
    (the guard in q_driver_post / parse_round)."""
    import ctypes as C
    from simharness import TABLES
    data = G.enwik_text(5000, seed=3, vocab=500)
    for quality in (5, 9):
        sizes = (C.c_uint64 * 1)(len(data))
        ops = (C.c_uint8 * 1)(7)
        out = C.create_string_buffer(20000)
        assert sim.L.sim_stream(TABLES.encode(), data, len(data), quality, 22, 1 << 20, 0, sizes, ops, 1, 0, out, 20000) == -3
