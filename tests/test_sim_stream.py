"""The tiled parse of ONE unpartitioned quality-5 stream longer than the window (JOB_FLAG_STREAMT: k_tile.h stream_*,
k_chain.h in stream positions, index chunks with a look-back) on the host SIMT simulator, byte for byte against the
reference library's one-shot BrotliEncoderCompress — several laps of the ring buffer at lgwin 17 / 18, meta-block cuts,
wraps of the 16-bit store counter — and through brotli_amd/csrc/encode_abi.c (the stock call, routed to the tiled path
and, where the stream does not suit the tiles, on to the serial device stream).  The `-m gpu` tests run the same at
full size through the C ABI."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import gen_inputs as G
from simharness import Sim

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import fuzz_stream_sim  # noqa: E402


@pytest.fixture(scope="module")
def sim():
    return Sim()


def _same(sim, ref, data, lgwin, **kw):
    got, info = sim.encode_stream(data, lgwin=lgwin, **kw)
    assert got is not None, "left the tiled path: reasons %#x" % info[0]
    assert got == ref.compress(data, 5, lgwin)
    return info


@pytest.mark.parametrize("n,lgwin", [(1 << 30, 22), ((1 << 31) - 1, 22), (1 << 30, 17), (123456789, 19), (5 << 20, 22)])
def test_workspace_arithmetic_at_sizes_the_simulator_cannot_run(sim, n, lgwin):
    """plan_stream + stream_emit_mb (the meta-blocks' workspace, carved out of stream-sized regions by position on the
    device) for a GiB and for the largest stream the path takes: every meta-block's regions inside the stream's,
    none overlapping its neighbour's, for random cut patterns with the most commands a meta-block can hold."""
    sim.L.sim_stream_layout_check.restype = C.c_long
    sim.L.sim_stream_layout_check.argtypes = [C.c_uint64, C.c_int, C.c_uint32]
    for seed in range(6):
        assert sim.L.sim_stream_layout_check(n, lgwin, seed) == 0


@pytest.mark.parametrize("n,lgwin,seed", [(250000, 17, 1), (131073, 17, 7), (430000, 17, 2)])
def test_text_streams(sim, ref, n, lgwin, seed):
    """Two tiles and a byte; inside one lap of the ring; 1.6 laps (stale bytes behind a block end, candidates at
    the ring's physical end, window ageing); two or three meta-blocks each."""
    info = _same(sim, ref, bytes(G.enwik_text(n, seed=seed)), lgwin, reverse=seed & 1)
    assert info[2] >= (2 if n > 200000 else 1)


@pytest.mark.parametrize("n,lgwin,seed,kind", [(430000, 17, 2, "text"), (600000, 18, 3, "text"), (500000, 17, 4, "mix"),
                                                (250000, 23, 1, "text"), (330000, 24, 2, "text")])
def test_chunks_of_half_a_window(sim, ref, monkeypatch, n, lgwin, seed, kind):
    """lgwin 24 beyond 16 MiB — what the reference's CLI chooses for every big file — has index chunks of HALF a window, so
    that a chunk with its look-back stays within the 24-bit positions of an index entry (host_plan.h plan_stream): a search
    with fewer than 16 same-key entries before it in its chunk is exact and goes on in the chunk before (k_index.h
    IxGeom::older, k_chain.h c_search_exact), a changed store bit raises events two chunks on (k_tile.h stream_events).
    BROTLI_AMD_HALF_CHUNKS=1 splits every window that way: at lgwin 17 a chunk is ONE tile and most searches take the new
    path.  (The last two cases: lgwin 23 / 24 themselves, streams inside one chunk.)  The `-m gpu` suite runs 40 MiB ... 1 GiB
    at lgwin 24."""
    if lgwin < 23:
        monkeypatch.setenv("BROTLI_AMD_HALF_CHUNKS", "1")
    data = bytes(G.enwik_text(n, seed=seed)) if kind == "text" else bytes(G.mixed_corpus(n, seed=seed))
    got, info = sim.encode_stream(data, lgwin=lgwin, reverse=seed & 1)
    if got is None and kind == "mix":
        pytest.skip("the mix left the tiled path (reasons %#x): nothing to compare" % info[0])
    assert got is not None and got == ref.compress(data, 5, lgwin)


@pytest.mark.parametrize("seed", [31042, 31054])
def test_half_chunks_changed_stretch_reaches_two_chunks_on(sim, ref, monkeypatch, seed):
    """tools/fuzz_stream_sim.py with BROTLI_AMD_HALF_CHUNKS=1, seeds 31042 / 31054: a changed, unstored position that
    leaves its successor walk to the changed entry before it (k_tile.h stream_events) — the walk into the chunk after
    next ended where the window of the FIRST entry of such a stretch ends, though the last one's reaches further."""
    monkeypatch.setenv("BROTLI_AMD_HALF_CHUNKS", "1")
    data, lgwin, kind = fuzz_stream_sim.make(seed)
    got, info = sim.encode_stream(data, lgwin=lgwin, reverse=seed & 1)
    assert got is not None and got == ref.compress(data, 5, lgwin)


def test_half_chunks_with_a_counter_wrap(sim, ref, monkeypatch):
    """The 16-bit store counter's zones (k_stream_zones) over chunks of half a window: the stream of
    test_counter_wrap_changes_the_bytes_and_is_followed."""
    monkeypatch.setenv("BROTLI_AMD_HALF_CHUNKS", "1")
    rng = np.random.default_rng(1)
    words = [bytes(rng.integers(97, 123, int(rng.integers(5, 10)), dtype=np.uint8)) + b" " for _ in range(2)]
    n = 1300000
    data = (bytes(rng.integers(97, 123, 30000, dtype=np.uint8)) + b"".join(words[i] for i in rng.integers(0, 2, n // 5)))[:n]
    got, info = sim.encode_stream(data, lgwin=17)
    assert got == ref.compress(data, 5, 17)


def test_counter_wrap_changes_the_bytes_and_is_followed(sim, ref, monkeypatch):
    """Two words in random order behind 30 kB of noise (which closes the static-dictionary gate): every inner
    4-byte key is stored more than 65536 times, and the first searches behind a wrap see fewer ring slots
    (hash_longest_match_simd_inc.h: num_ as uint16).  Without the marks of k_stream_zones the stream comes out wrong."""
    rng = np.random.default_rng(1)
    words = [bytes(rng.integers(97, 123, int(rng.integers(5, 10)), dtype=np.uint8)) + b" " for _ in range(2)]
    n = 1300000
    data = (bytes(rng.integers(97, 123, 30000, dtype=np.uint8)) + b"".join(words[i] for i in rng.integers(0, 2, n // 5)))[:n]
    want = ref.compress(data, 5, 17)
    got, info = sim.encode_stream(data, lgwin=17)
    assert got == want
    monkeypatch.setenv("SIM_NO_ZONES", "1")
    got, info = sim.encode_stream(data, lgwin=17)
    assert got is not None and got != want


def test_copies_cut_by_block_ends(sim, ref):
    """Found by tools/fuzz_stream_sim.py (seed 15): a copy of the chain's fast path that ran to its block's end marked
    the block's last three positions as unstored, though the next block's stitch stores them — the plain chain had
    the same fault.  Both are checked here."""
    data, lgwin, kind = fuzz_stream_sim.make(15)
    assert kind == 3
    data = data[:540000]        # (the position that came out wrong is 478 843, its missing candidate 458 750)
    _same(sim, ref, data, lgwin, reverse=1)
    assert sim.encode(data, 5, 22, flags=64) == ref.compress(data, 5, 22)


def test_copies_longer_than_a_block(sim, ref):
    """A 150 kB piece that comes again twice: ExtendLastCommand consumes whole input blocks (tiles without a command
    and without pending literals), the command it lengthens belongs to a tile two or three blocks back."""
    text = bytes(G.enwik_text(500000, seed=5))
    piece = text[100000:250000]
    data = text[:300000] + piece + text[300000:380000] + piece + piece[:70000] + text[380000:]
    info = _same(sim, ref, data, 19)
    assert info[2] >= 1


def test_streams_that_leave_the_tiled_path(sim, ref):
    """Random bytes (most positions unstored by the literal spree): the tiled path says so and writes nothing — the
    library then runs the serial device stream.  The mixed corpus (floats, sparse zeros, noise, text) stays on it."""
    rng = np.random.default_rng(9)
    got, info = sim.encode_stream(bytes(rng.integers(0, 256, 300000, dtype=np.uint8)), lgwin=17)
    assert got is None and info[0] & 0x8000
    _same(sim, ref, bytes(G.mixed_corpus(262144, seed=6)), 17)


def test_a_raw_meta_block_rolls_the_distance_cache_back(sim, ref):
    """400 kB of random bytes inside text: one meta-block of the stream is stored uncompressed (encode.c:598-614) — its
    payload byte aligned in the stream, and the tile behind it starts from the distance cache the raw meta-block
    started from: k_stream_scan decides which are raw, k_stream_rollback tells the tiles, the sweep loop runs again."""
    rng = np.random.default_rng(9)
    text = bytes(G.enwik_text(700000, seed=3))
    data = text[:300000] + bytes(rng.integers(0, 256, 400000, dtype=np.uint8)) + text[300000:600000]
    want = ref.compress(data, 5, 17)
    assert len(want) > 400000           # (the noise did not compress)
    got, info = sim.encode_stream(data, lgwin=17)
    assert got == want and info[2] >= 5


def test_the_dictionary_gate_stays_open(sim, ref):
    """Data on which the static dictionary's gate never closes (hash.h:186) — English, where the dictionary keeps
    matching, and a highly repetitive stream, where hardly a search fails: tile 0 ends with the gate open, the other
    tiles start over with the gate taken as open for good, and the summed counters confirm it (k_tile.h)."""
    alice = open(os.path.join(HERE, "golden", "alice29.txt"), "rb").read()
    _same(sim, ref, (alice * 3)[:400000], 17)
    rng = np.random.default_rng(3)
    words = [bytes(rng.integers(97, 123, 7, dtype=np.uint8)) + b" " for _ in range(4)]
    _same(sim, ref, b"".join(words[i] for i in rng.integers(0, 4, 60000))[:300000], 17, reverse=1)


@pytest.mark.parametrize("seed,n", [(5004, 400000), (6000, 400000)])
def test_the_dictionary_gate_closes_in_a_later_block(sim, ref, seed, n):
    """English, then text the dictionary does not match: the gate closes somewhere behind the first block.  The walk
    over the summed counters (k_tile.h: gate_walk) finds the tile in which it may close, that tile is parsed again
    from the exact counters, and everything behind the tile that ends closed is parsed again as closed.  (Seed 6000
    is the stream that showed the replay counting a command's lookups twice.)"""
    data, lgwin, kind = fuzz_stream_sim.make(seed)
    assert kind == 5
    _same(sim, ref, data[:n], lgwin)


@pytest.mark.parametrize("seed", [300, 301, 302, 303, 10000, 10001, 10002])
def test_fuzz_slice(sim, ref, seed):
    data, lgwin, kind = fuzz_stream_sim.make(seed)
    data = data[:400000]
    got, info = sim.encode_stream(data, lgwin=lgwin, reverse=seed & 1)
    if got is not None:
        assert got == ref.compress(data, 5, lgwin)
    else:
        assert info[0] != 0


def test_stock_call_through_the_boundary(ref, monkeypatch):
    """BrotliEncoderCompress(5, lgwin, ...) of encode_abi.c over the simulator: a text longer than the window takes the
    tiled stream path, a stream that does not suit it the serial device stream — the bytes are the reference's either
    way; BROTLI_AMD_STREAM_TILES=0 keeps everything on the serial stream."""
    from test_abi_on_sim import SIM_ABI
    from test_gpu_abi import _bind
    from refharness import ROOT, TABLES
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "simt")], check=True)
    monkeypatch.setenv("BROTLI_AMD_TABLES", TABLES)
    L = _bind(SIM_ABI)

    def call(data, lgwin):
        cap = L.BrotliEncoderMaxCompressedSize(len(data))
        out = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        assert L.BrotliEncoderCompress(5, lgwin, 0, len(data), data, C.byref(n), out)
        return out.raw[:n.value]

    text = bytes(G.enwik_text(280000, seed=12))
    assert call(text, 17) == ref.compress(text, 5, 17)
    mixed = bytes(G.mixed_corpus(200000, seed=5))
    assert call(mixed, 17) == ref.compress(mixed, 5, 17)
    # input handed over in pieces: with BROTLI_AMD_FEED_KB above the input's size the library holds it until FINISH,
    # and the FINISH takes the tiled stream path like the one-shot call
    from test_gpu_abi import drive
    monkeypatch.setenv("BROTLI_AMD_FEED_KB", "1000000")
    got, fin = drive(L, text, [(100000, 0), (100001, 0), (len(text) - 200001, 2)], params=((2, 17),))
    assert fin and bytes(got) == ref.compress(text, 5, 17)
    monkeypatch.delenv("BROTLI_AMD_FEED_KB")
    # ... and without any vendor setting when the caller announced the size (BROTLI_PARAM_SIZE_HINT = 5: what the CLI
    # does for a file), next to the reference library driven the same way
    from test_gpu_abi import _bind as bind2
    stock = bind2(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
    ops = [(65536, 0)] * 3 + [(len(text) - 3 * 65536, 2)]
    got, fin = drive(L, text, ops, params=((2, 17), (5, len(text))))
    want, fin2 = drive(stock, text, ops, params=((2, 17), (5, len(text))))
    assert fin and fin2 and bytes(got) == bytes(want)
    # what the tiled path does not take stays where it was: a stream offset (BROTLI_PARAM_STREAM_OFFSET = 9), an empty
    # FLUSH in front (the header gone: BROTLI_AMD_FLAG_NO_HEADER), literal context modelling switched off (4)
    for ops, params in (([(len(text), 2)], ((2, 17), (9, 70000))), ([(0, 1), (len(text), 2)], ((2, 17),)),
                        ([(len(text), 2)], ((2, 17), (4, 1)))):
        got, fin = drive(L, text, ops, params=params)
        want, fin2 = drive(stock, text, ops, params=params)
        assert fin and fin2 and bytes(got) == bytes(want)
    monkeypatch.setenv("BROTLI_AMD_STREAM_TILES", "0")
    assert call(text[:200000], 17) == ref.compress(text[:200000], 5, 17)


def test_stock_call_at_the_windows_the_cli_chooses(ref, monkeypatch):
    """encode_abi.c's routing at lgwin 23 / 24 (c/tools/brotli.c:1434-1447 picks them for files above 4 MiB): an input
    that the one-shard job does not take (above 4 MiB - 16) goes to the tiled stream; the bytes are the reference's.
    BROTLI_AMD_VERBOSE would say so if it had gone to the serial stream (minutes on the simulator for this size)."""
    from test_abi_on_sim import SIM_ABI
    from test_gpu_abi import _bind
    from refharness import ROOT, TABLES
    import subprocess
    import time
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "simt")], check=True)
    monkeypatch.setenv("BROTLI_AMD_TABLES", TABLES)
    L = _bind(SIM_ABI)
    data = bytes(G.enwik_text((4 << 20) + 70000, seed=44))
    for lgwin in (23, 24):
        cap = L.BrotliEncoderMaxCompressedSize(len(data))
        out = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        t0 = time.time()
        assert L.BrotliEncoderCompress(5, lgwin, 0, len(data), data, C.byref(n), out)
        assert out.raw[:n.value] == ref.compress(data, 5, lgwin)
        assert time.time() - t0 < 400


@pytest.mark.parametrize("lgwin,blocks", [(17, 1), (17, 2), (17, 3), (17, 4), (17, 7), (17, 8), (18, 2), (18, 3), (18, 9)])
def test_process_calls_ending_on_a_block_boundary_then_an_empty_finish(ref, lgwin, blocks):
    """PROCESS calls that fill the last input block exactly, then FINISH with nothing: the reference has encoded that
    block with is_last = 0 already (encode.c:1700-1712) — when the rule of encode.c:1141-1166 closes the meta-block
    there, it leaves with ISLAST = 0 and an empty last meta-block follows, which is NOT the one-shot stream (3 and 7
    blocks of this text at lgwin 17).  The library holds such input for one job (with or without an announced size):
    the tiled stream with BROTLI_AMD_FLAG_TAIL_FINISH (host_plan.h: stream_tail_fix), a one-shard job where the rule
    cannot apply, the serial stream fed the same way otherwise."""
    from test_abi_on_sim import SIM_ABI
    from test_gpu_abi import _bind, drive
    from refharness import ROOT, TABLES
    os.environ["BROTLI_AMD_TABLES"] = TABLES
    L = _bind(SIM_ABI)
    stock = _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
    text = bytes(G.enwik_text(blocks * 65536, seed=12))
    for hint in (0, len(text)):
        params = ((2, lgwin),) + (((5, hint),) if hint else ())
        ops = [(65536, 0)] * blocks + [(0, 2)]
        got, fin = drive(L, text, ops, params=params)
        want, fin2 = drive(stock, text, ops, params=params)
        assert fin and fin2 and bytes(got) == bytes(want), (lgwin, blocks, hint)


@pytest.mark.parametrize("quality,lgwin,block", [(2, 18, 16384), (3, 20, 16384), (4, 18, 65536), (6, 18, 65536), (9, 17, 1 << 18), (7, 16, 65536), (5, 16, 65536)])
def test_empty_finish_behind_complete_blocks_at_the_other_qualities(ref, quality, lgwin, block):
    """The same call sequence at qualities 2 - 4 and 6 - 9 and at the small windows: one-shard jobs where the rule
    that closes a meta-block cannot apply, the serial device stream fed as the caller fed the library otherwise."""
    from test_abi_on_sim import SIM_ABI
    from test_gpu_abi import _bind, drive
    from refharness import ROOT, TABLES
    os.environ["BROTLI_AMD_TABLES"] = TABLES
    L = _bind(SIM_ABI)
    stock = _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
    for k in (1, 2, 3, 5, 8):
        text = bytes(G.enwik_text(k * block, seed=12 + k))
        params = ((1, quality), (2, lgwin))
        ops = [(block, 0)] * k + [(0, 2)]
        got, fin = drive(L, text, ops, params=params)
        want, fin2 = drive(stock, text, ops, params=params)
        assert fin and fin2 and bytes(got) == bytes(want), (quality, lgwin, k)


@pytest.mark.parametrize("quality,lgwin,feed_kb,ops", [
    (9, 17, 320, [(327680, 0), (196608, 0), (0, 2)]),        # forwarded 320 KiB (not a multiple of the 256 KiB block), 512 KiB in all
    (9, 17, 320, [(327680, 0), (196608 + 1000, 0), (0, 2)]),  # the same with an incomplete last block
    (6, 18, 100, [(102400, 0), (28672, 0), (0, 2)]),           # 100 KiB forwarded, 128 KiB = two blocks in all
])
def test_empty_finish_behind_an_unaligned_forward(ref, monkeypatch, quality, lgwin, feed_kb, ops):
    """PROCESS forwarded part of the input before the empty FINISH (BROTLI_AMD_FEED_KB), and what is still held is not
    a whole number of blocks although the stream is: the reference counts its blocks from the stream's start
    (encode.c:1700-1712), so the last one was encoded with is_last = 0 and the FINISH adds an empty meta-block — the
    serial stream is fed the calls as they came (the alignment test looked at the held bytes only)."""
    from test_abi_on_sim import SIM_ABI
    from test_gpu_abi import _bind, drive
    from refharness import ROOT, TABLES
    os.environ["BROTLI_AMD_TABLES"] = TABLES
    monkeypatch.setenv("BROTLI_AMD_FEED_KB", str(feed_kb))
    L = _bind(SIM_ABI)
    stock = _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))
    text = bytes(G.enwik_text(sum(n for n, _ in ops), seed=31))
    params = ((1, quality), (2, lgwin))
    got, fin = drive(L, text, ops, params=params)
    want, fin2 = drive(stock, text, ops, params=params)
    assert fin and fin2 and bytes(got) == bytes(want)


@pytest.mark.parametrize("reverse", [0, 1])
def test_every_big_bucket_on_the_lists_of_k_ix_big(sim, ref, monkeypatch, reverse):
    """BROTLI_AMD_IX_GIANT=320: every bucket too big for LDS goes onto the block lists and is searched by k_ix_big /
    k_ix_big_s (by default only the ones above 2048 entries do — a run of zeros —, the others are walked by the wave that
    sorted them): a stream with a raw meta-block, and a plan of two long shards, in both lane orders."""
    monkeypatch.setenv("BROTLI_AMD_IX_GIANT", "320")
    rng = np.random.default_rng(9)
    text = bytes(G.enwik_text(500000, seed=3))
    data = text[:200000] + bytes(rng.integers(0, 256, 150000, dtype=np.uint8)) + bytes(200000) + text[200000:420000]
    got, info = sim.encode_stream(data, lgwin=17, reverse=reverse)
    assert got == ref.compress(data, 5, 17)
    from refharness import Oracle
    from test_sim_kernels import IX_LAYOUTS, _oracle_plan
    from simharness import Sim
    plan = bytes(G.enwik_text(320 << 10, seed=21, vocab=4000))
    want = _oracle_plan(Oracle(), plan, 1 << 30, 160 << 10)
    assert Sim().encode(plan, 5, 22, 1 << 30, 160 << 10, flags=IX_LAYOUTS["groups4"], reverse=reverse) == want
