"""Pins the C restatement (oracle/brotli_oracle.c) against the reference
encoder itself (oracle/_ref) and against the committed golden vectors.

The reference's own tests hold no encoder-byte fixtures, so this
is the pin: same inputs, same parameters, same partition plan -> same bytes.
"""
import hashlib
import json
import os
import sys

import pytest

import gen_inputs as G

HERE = os.path.dirname(os.path.abspath(__file__))
ALICE = os.path.join(HERE, "golden", "alice29.txt")


def _inputs():
    text = G.enwik_text(1 << 20, seed=11, vocab=20000)
    yield "text1m", text
    yield "rand64k", G.random_bytes(1 << 16)
    yield "mixed1m", G.mixed_corpus(1 << 20)
    yield "zeros", bytes(300000)
    yield "rle", (b"abcdefgh" * 50000)[:333333]
    yield "tiny1", b"x"
    yield "tiny2", b"xy"
    yield "tiny3", b"xyz"
    yield "tiny9", b"123456789"
    yield "64x", b"x" * 64
    yield "text_rand", text[:200000] + G.random_bytes(150000) + text[:100000]


INPUTS = dict(_inputs())


@pytest.mark.parametrize("name", list(INPUTS))
@pytest.mark.parametrize("quality,lgwin", [(5, 22), (6, 22), (5, 18), (5, 24),
                                           (7, 22), (9, 24), (8, 17)])
def test_mode_s_matches_reference(ref, oracle, name, quality, lgwin):
    data = INPUTS[name]
    assert oracle.encode_plan(data, quality, lgwin, 0) == \
        ref.compress(data, quality, lgwin)


@pytest.mark.parametrize("name", ["text1m", "mixed1m", "zeros", "rle", "tiny3", "tiny9", "64x", "text_rand"])
@pytest.mark.parametrize("quality,lgwin", [(2, 22), (3, 22), (4, 22), (4, 18), (3, 16), (2, 10), (4, 24)])
def test_qualities_2_to_4_match_reference(ref, oracle, name, quality, lgwin):
    """The HashLongestMatchQuickly family (H2, H3, H4; H54 once a MiB is announced) with the
    trivial / fast / greedy meta-block writers: one stream, and shards with STREAM_OFFSET."""
    data = INPUTS[name]
    assert oracle.encode_plan(data, quality, lgwin, 0) == ref.compress(data, quality, lgwin)
    if len(data) > 200000:
        assert oracle.encode_plan(data, quality, lgwin, 100000) == ref.encode_plan(data, quality, lgwin, 100000)
        # a small announced size: H4 instead of H54 at quality 4
        part = data[:150000]
        assert oracle.encode_shard(part, quality, lgwin, len(part), 70000, False) == \
            ref.encode_shard(part, quality, lgwin, len(part), 70000, False)


@pytest.mark.parametrize("name", ["text1m", "mixed1m", "zeros", "rle", "tiny9", "text_rand"])
@pytest.mark.parametrize("quality,lgwin", [(5, 16), (6, 14), (7, 16), (8, 12), (9, 16), (9, 10)])
def test_small_windows_forgetful_chain_match_reference(ref, oracle, name, quality, lgwin):
    """lgwin <= 16 at qualities 5 - 9: the forgetful-chain hashers H40 / H41 / H42 (chains through
    banks of recycled slots, tiny-hash filter on the distance cache); inputs lap the 128 KiB ring."""
    data = INPUTS[name]
    assert oracle.encode_plan(data, quality, lgwin, 0) == ref.compress(data, quality, lgwin)
    if len(data) > 200000:
        assert oracle.encode_plan(data, quality, lgwin, 100000) == ref.encode_plan(data, quality, lgwin, 100000)


@pytest.mark.parametrize("nbytes,dict_bytes,nchunks", [(50000, 20000, 1), (300000, 300000, 3), (3000, 100000, 1),
                                                       (70000, 9, 1)])
@pytest.mark.parametrize("quality,lgwin", [(5, 22), (5, 18), (6, 22), (9, 24), (7, 17), (3, 22), (4, 18), (2, 22),
                                           (5, 16), (7, 14), (9, 12)])
def test_attached_dictionaries_match_reference(ref, oracle, quality, lgwin, nbytes, dict_bytes, nchunks):
    """BrotliEncoderPrepareDictionary(RAW) + AttachPreparedDictionary (encode.h:318-363): the lookup
    after every FindLongestMatch (hash.h:526-717), `gap` in the distance code and the static-dictionary
    distances, ExtendLastCommand into the dictionary; H2 only gets the gap (no compound variant)."""
    data, chunks = G.dictionary_case(nbytes, dict_bytes, nchunks, seed=quality * 100 + lgwin)
    want = ref.encode_calls(data, quality, lgwin, [(len(data), 2)], dictionaries=chunks)
    oracle.set_dictionary(chunks)
    try:
        got = oracle.encode_shard(data, quality, lgwin, 0, 0, True)
    finally:
        oracle.set_dictionary(())
    assert got == want
    if quality != 2 and dict_bytes > 1000:
        assert len(want) < len(ref.compress(data, quality, lgwin))
    assert ref.decompress_with(want, len(data), chunks) == data


def test_attached_dictionary_large_hint_h54_and_h68(ref, oracle):
    """Past 1 MiB: H54 at quality 4 (plain variant, gap only), H68 at quality 5."""
    data, chunks = G.dictionary_case((1 << 20) + 50000, 150000, 2, seed=77)
    for quality in (4, 5):
        want = ref.encode_calls(data, quality, 22, [(len(data), 2)], dictionaries=chunks)
        oracle.set_dictionary(chunks)
        try:
            got = oracle.encode_shard(data, quality, 22, 0, 0, True)
        finally:
            oracle.set_dictionary(())
        assert got == want


@pytest.mark.parametrize("name", ["text1m", "mixed1m", "rle", "text_rand"])
@pytest.mark.parametrize("shard", [1 << 18, 100000, 65536, 70000])
def test_mode_p_matches_reference(ref, oracle, name, shard):
    data = INPUTS[name]
    want = ref.encode_plan(data, 5, 22, shard)
    assert oracle.encode_plan(data, 5, 22, shard) == want
    assert ref.decompress(want, len(data)) == data


@pytest.mark.parametrize("quality,lgwin,lgblock", [(5, 22, 17), (5, 18, 20), (9, 22, 16), (6, 20, 21), (4, 22, 18), (5, 22, 12),
                                                   (7, 16, 24), (3, 20, 20)])
def test_lgblock_parameter_matches_reference(ref, oracle, quality, lgwin, lgblock):
    """BROTLI_PARAM_LGBLOCK (encode.h:190-197; ComputeLgBlock, quality.h:75-92): the restatement with the block size set
    (oracle_set_lgblock) equals the reference with the parameter set — whole streams and shards with stream offsets."""
    data = INPUTS["text_rand"] + INPUTS["mixed1m"][:300000]
    hint = 1 << 20
    try:
        oracle.set_lgblock(lgblock)
        for off, n, last in ((0, len(data), True), (0, 200000, False), (200000, 300001, False), (500001, len(data) - 500001, True)):
            piece = data[off:off + n]
            want = ref.encode_shard(piece, quality, lgwin, hint, off, last, lgblock=lgblock)
            assert oracle.encode_shard(piece, quality, lgwin, hint, off, last) == want, (off, n)
    finally:
        oracle.set_lgblock(0)


def test_large_hint_small_shards(ref, oracle):
    """H68 (size_hint >= 1 MiB) with shards far below 1 MiB, ragged tail,
    shards of 1, 2 and 3 bytes (flint edge cases, encode.c:1667-1708)."""
    data = G.enwik_text((1 << 20) + 3, seed=3, vocab=20000)
    for shard in (1 << 17, (1 << 20) + 1, (1 << 20) + 2, 1 << 20):
        assert oracle.encode_plan(data, 5, 22, shard) == \
            ref.encode_plan(data, 5, 22, shard)


def test_multi_metablock_and_ring_wrap(ref, oracle):
    """> 16 MiB: several meta-blocks and a second lap of the 8 MiB ring."""
    data = G.enwik_text(18 << 20, seed=5, vocab=20000)
    assert oracle.encode_plan(data, 5, 22, 0) == ref.compress(data, 5, 22)


def test_alice29_known_answer(oracle):
    """Known answer recorded in SURVEY.md §6 / BASELINE.md §2 from the
    reference CLI and Python binding: 152089 -> 52809 bytes."""
    data = open(ALICE, "rb").read()
    out = oracle.encode_plan(data, 5, 22, 0)
    assert len(out) == 52809
    assert hashlib.sha256(out).hexdigest() == \
        "b4bf4f4f62af5e94769b822b0f24e4edebdb95477e001b4886eb02f2043834f3"


def test_golden_vectors(oracle):
    """Committed fixtures generated by tests/golden/make_golden.py from the
    reference library (sha256 of the .br for seeded inputs)."""
    gold = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    for case in gold["cases"]:
        data = G.make(case["input"])
        out = oracle.encode_plan(data, case["quality"], case["lgwin"],
                                 case["shard_size"])
        assert len(out) == case["size"], case
        assert hashlib.sha256(out).hexdigest() == case["sha256"], case


def test_golden_vectors_lgblock(oracle):
    """The golden vectors with BROTLI_PARAM_LGBLOCK set (tests/golden/make_golden.py: LGBLOCK_CASES)."""
    golden = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    try:
        for c in golden["lgblock_cases"]:
            data = G.make(c["input"])
            oracle.set_lgblock(c["lgblock"])
            out = oracle.encode_plan(data, c["quality"], c["lgwin"], c["shard_size"])
            assert len(out) == c["size"] and hashlib.sha256(out).hexdigest() == c["sha256"], c
    finally:
        oracle.set_lgblock(0)


def test_golden_vectors_dictionaries(oracle):
    gold = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    for case in gold["dictionary_cases"]:
        data, chunks = G.dictionary_case(**case["input"])
        oracle.set_dictionary(chunks)
        try:
            if case.get("shard_size"):
                out = oracle.encode_plan(data, case["quality"], case["lgwin"], case["shard_size"])
            else:
                out = oracle.encode_shard(data, case["quality"], case["lgwin"], 0, 0, True)
        finally:
            oracle.set_dictionary(())
        assert len(out) == case["size"] and hashlib.sha256(out).hexdigest() == case["sha256"], case


def test_golden_vectors_quality1(oracle):
    """The quality-1 fixtures (one-shot and CLI-style feeds) of tests/golden/golden.json."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import q1_calls
    gold = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    assert gold["quality1_cases"]
    for case in gold["quality1_cases"]:
        data = G.make(case["input"])
        out = oracle.encode_fast(data, case["lgwin"], q1_calls(len(data), case["feed_kb"]))
        assert len(out) == case["size"], case
        assert hashlib.sha256(out).hexdigest() == case["sha256"], case


# --- quality 1: the two-pass fragment compressor -------------------------

def _calls(n, chunk, flush_mid=False, empty_finish=False):
    """CLI-style call sequence: `chunk` bytes per PROCESS call, FINISH with the last one
    (or as an extra empty call, which is what the CLI does when the size is a multiple
    of its read size, c/tools/brotli.c:1419-1463)."""
    from refharness import OP_FINISH, OP_FLUSH, OP_PROCESS
    calls, off = [], 0
    while off < n:
        m = min(chunk, n - off)
        off += m
        calls.append((m, OP_FINISH if off == n and not empty_finish else OP_PROCESS))
    if flush_mid and len(calls) > 1:
        calls[len(calls) // 2] = (calls[len(calls) // 2][0], OP_FLUSH)
    if empty_finish or not calls:
        calls.append((0, OP_FINISH))
    return calls


Q1_INPUTS = dict(INPUTS)
Q1_INPUTS["text3m+"] = G.enwik_text((3 << 20) + 12345, seed=7, vocab=20000)
for _n in (15, 16, 17, 255, 257, 32768, 32769, 131072, 131073):
    Q1_INPUTS["text%d" % _n] = G.enwik_text(_n, seed=_n, vocab=2000)


@pytest.mark.parametrize("name", list(Q1_INPUTS))
def test_q1_one_call_matches_reference(ref, oracle, name):
    """BrotliEncoderCompressStream(FINISH) with the whole input: fragments of 1 << lgwin,
    table size / min-match chosen from the fragment size (encode.c:148-189, 1478)."""
    from refharness import OP_FINISH
    data = Q1_INPUTS[name]
    for lgwin in (10, 16, 18, 22, 24):
        assert oracle.encode_fast(data, lgwin) == ref.encode_calls(data, 1, lgwin, [(len(data), OP_FINISH)]), lgwin


@pytest.mark.parametrize("name", ["text1m", "mixed1m", "text_rand", "text3m+", "rle", "tiny3"])
@pytest.mark.parametrize("chunk", [65536, 100000, 524288])
def test_q1_call_sequences_match_reference(ref, oracle, name, chunk):
    data = Q1_INPUTS[name]
    for kw in ({}, {"flush_mid": True, "empty_finish": True}):
        calls = _calls(len(data), chunk, **kw)
        want = ref.encode_calls(data, 1, 22, calls)
        assert oracle.encode_fast(data, 22, calls) == want
        assert ref.decompress(want, len(data)) == data


def test_q1_random_bytes_become_one_raw_meta_block_per_fragment(ref, oracle):
    """BASELINE config 3 in miniature: every 128 KiB block fails ShouldCompress and the
    fragment is rewritten as a single uncompressed meta-block (:622-627)."""
    from refharness import OP_FINISH
    data = G.random_bytes(3 << 20, seed=5)
    got = oracle.encode_fast(data, 20)
    assert got == ref.encode_calls(data, 1, 20, [(len(data), OP_FINISH)])
    assert len(got) <= len(data) + 3 * 5 + 2
