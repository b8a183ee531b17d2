"""Parity tests proper: the HIP path (through the C ABI of
brotli_amd/lib/libbrotli_amd_hip.so) against the oracle on the same inputs —
bit-exact — and, at BASELINE.json sizes, through size-independent properties
(decode round trip with the reference decoder, per-shard independence)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import gen_inputs as G

# a kernel that never returns must not take the whole GPU tier with it: pytest-timeout's
# thread method ends the run (a blocked HIP call cannot be interrupted by a signal)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]
HERE = os.path.dirname(os.path.abspath(__file__))
ALICE = open(os.path.join(HERE, "golden", "alice29.txt"), "rb").read()


@pytest.fixture(scope="module")
def ctx():
    import torch
    torch.cuda.init()       # same initialisation order as bench.py: torch first
    from brotli_amd import hip
    c = hip.Context(0)      # raises if the HIP library or a gfx950 device is missing
    yield c
    c.close()


def _gpu(ctx, data, shard=0, hint=0, quality=5, lgwin=22, **kw):
    from brotli_amd import hip
    out, info = ctx.encode_host(data, hip.make_params(quality, lgwin, shard, hint, **kw))
    return out, info


def _oracle_plan(oracle, data, shard, hint=0, base=0, is_last=True):
    n = len(data)
    shard = shard or n
    hint = hint or min(base + n, 1 << 30)
    parts, off = [], 0
    while off < n:
        m = min(shard, n - off)
        parts.append(oracle.encode_shard(data[off:off + m], 5, 22, hint, min(base + off, 1 << 30),
                                         is_last and off + m == n))
        off += m
    return b"".join(parts)


TEXT4M = G.enwik_text(4 << 20, seed=11, vocab=20000)
CASES = {
    "alice": (ALICE, 0),
    "alice_8shards": (ALICE, 20000),
    "tiny1": (b"x", 0), "tiny2": (b"xy", 0), "tiny3": (b"xyz", 0), "tiny9": (b"123456789", 0),
    "x64": (b"x" * 64, 0),
    "shards_of_1_2_3": (b"abcabcabcabc", 3),
    "zeros": (bytes(300000), 0),
    "rle": ((b"abcdefgh" * 50000)[:333333], 0),
    "random_raw": (G.random_bytes(1 << 16), 0),
    "random_1m_64k": (G.random_bytes(1 << 20), 1 << 16),
    "text4m_single_stream": (TEXT4M, 0),
    "text4m_256k": (TEXT4M, 1 << 18),
    "text4m_ragged_100000": (TEXT4M, 100000),
    "text4m_64k": (TEXT4M, 1 << 16),
    "mixed2m_128k": (G.mixed_corpus(2 << 20), 1 << 17),
    "text_rand_text_64k": (TEXT4M[:200000] + G.random_bytes(150000) + TEXT4M[:100000], 1 << 16),
    "text_rand_text_single": (TEXT4M[:200000] + G.random_bytes(150000) + TEXT4M[:100000], 0),
    "small_hint_h58": (TEXT4M[:900000], 1 << 17),      # < 1 MiB total: H58, simple context maps
}


@pytest.mark.parametrize("name", list(CASES))
def test_bytes_equal_oracle(ctx, oracle, name):
    data, shard = CASES[name]
    got, info = _gpu(ctx, data, shard)
    assert got == _oracle_plan(oracle, data, shard)
    assert info["out_bytes"] == len(got)


def test_alice_known_answer(ctx):
    """reference CLI / Python binding on alice29.txt, q5, lgwin 22."""
    got, _ = _gpu(ctx, ALICE)
    assert len(got) == 52809
    assert hashlib.sha256(got).hexdigest() == \
        "b4bf4f4f62af5e94769b822b0f24e4edebdb95477e001b4886eb02f2043834f3"


def test_golden_vectors(ctx):
    """Fixtures generated from the reference library by tests/golden/make_golden.py."""
    gold = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    n = 0
    for case in gold["cases"]:
        if case["quality"] not in range(2, 10) or case["lgwin"] not in range(10, 25):
            continue
        data = G.make(case["input"])
        if len(data) == 0:
            continue
        got, _ = _gpu(ctx, data, case["shard_size"], quality=case["quality"], lgwin=case["lgwin"])
        assert len(got) == case["size"], case
        assert hashlib.sha256(got).hexdigest() == case["sha256"], case
        n += 1
    assert n > 0


def test_golden_vectors_quality1(ctx):
    """Quality-1 fixtures from the reference library (one-shot and CLI-style feeds)."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import q1_calls
    gold = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    assert gold["quality1_cases"]
    for case in gold["quality1_cases"]:
        data = G.make(case["input"])
        calls = q1_calls(len(data), case["feed_kb"])
        got, nbits, _ = ctx.encode_fast_host(data, case["lgwin"], [c[0] for c in calls])
        assert nbits == 8 * len(got) and len(got) == case["size"], case
        assert hashlib.sha256(got).hexdigest() == case["sha256"], case


@pytest.mark.parametrize("lgwin", [18, 20, 24])
def test_other_windows(ctx, oracle, lgwin):
    from brotli_amd import hip
    data = TEXT4M[:(1 << 20) + 12345]
    got, _ = ctx.encode_host(data, hip.make_params(5, lgwin, 1 << 18))
    n = len(data)
    parts, off = [], 0
    while off < n:
        m = min(1 << 18, n - off)
        parts.append(oracle.encode_shard(data[off:off + m], 5, lgwin, n, off, off + m == n))
        off += m
    assert got == b"".join(parts)


@pytest.mark.parametrize("quality,lgwin,shard", [(2, 22, 1 << 17), (2, 18, 0), (3, 22, 1 << 18), (3, 16, 0), (4, 22, 1 << 17),
                                                 (4, 24, 0), (4, 10, 100000), (2, 10, 0)])
@pytest.mark.parametrize("name", ["text", "mixed", "text_rand_text", "zeros", "tiny"])
def test_qualities_2_to_4_equal_oracle(ctx, oracle, name, quality, lgwin, shard):
    """k_parse_quick.h (H2 / H3 / H4 / H54) with the single-block build and the trivial / fast
    meta-block writers: partition plans and single streams, windows from 10 to 24 bits."""
    from brotli_amd import hip
    data = {"text": TEXT4M[:(2 << 20) + 777], "mixed": G.mixed_corpus(2 << 20),
            "text_rand_text": TEXT4M[:200000] + G.random_bytes(150000) + TEXT4M[:100000],
            "zeros": bytes(300000), "tiny": b"hello hello hello hello"}[name]
    if len(data) < 1000 and shard:
        pytest.skip("one shard")
    got, info = ctx.encode_host(data, hip.make_params(quality, lgwin, shard))
    n = len(data)
    s = shard or n
    want = b"".join(oracle.encode_shard(data[o:o + s], quality, lgwin, min(n, 1 << 30), o, o + s >= n)
                    for o in range(0, n, s))
    assert got == want


@pytest.mark.parametrize("quality,lgwin,shard", [(5, 16, 1 << 17), (5, 10, 0), (6, 14, 0), (7, 16, 100000), (8, 12, 0),
                                                 (9, 16, 0), (9, 10, 1 << 16)])
@pytest.mark.parametrize("name", ["text", "mixed", "text_rand_text", "zeros"])
def test_small_windows_equal_oracle(ctx, oracle, name, quality, lgwin, shard):
    """Qualities 5 - 9 at lgwin 10 - 16: the forgetful-chain hashers H40 / H41 / H42."""
    from brotli_amd import hip
    data = {"text": TEXT4M[:(1 << 20) + 777], "mixed": G.mixed_corpus(1 << 20),
            "text_rand_text": TEXT4M[:200000] + G.random_bytes(150000) + TEXT4M[:100000],
            "zeros": bytes(300000)}[name]
    got, info = ctx.encode_host(data, hip.make_params(quality, lgwin, shard))
    n = len(data)
    s = shard or n
    want = b"".join(oracle.encode_shard(data[o:o + s], quality, lgwin, min(n, 1 << 30), o, o + s >= n)
                    for o in range(0, n, s))
    assert got == want


@pytest.mark.parametrize("quality,lgwin,shard", [(6, 22, 1 << 18), (7, 22, 1 << 18), (8, 20, 1 << 17),
                                                 (9, 22, 1 << 20), (9, 24, 0), (9, 24, 1 << 19)])
@pytest.mark.parametrize("name", ["text", "mixed", "text_rand_text"])
def test_deep_qualities_equal_oracle(ctx, oracle, name, quality, lgwin, shard):
    """BASELINE config 5 family: quality 9 / lgwin 24 (H6, 256-slot buckets, 16
    distance-cache probes, 256 KiB input blocks, up to 3 literal contexts) and
    qualities 6 - 8, single stream and partition plans."""
    from brotli_amd import hip
    data = {"text": TEXT4M, "mixed": G.mixed_corpus(2 << 20),
            "text_rand_text": TEXT4M[:200000] + G.random_bytes(150000) + TEXT4M[:100000]}[name]
    if shard == 0:
        data = data[:3 << 20]
    got, info = ctx.encode_host(data, hip.make_params(quality, lgwin, shard))
    n = len(data)
    s = shard or n
    want = b"".join(oracle.encode_shard(data[o:o + s], quality, lgwin, min(n, 1 << 30), o, o + s >= n)
                    for o in range(0, n, s))
    assert got == want


def test_multi_metablock_and_ring_wrap_single_stream(ctx, oracle):
    """One encoder instance over 18 MiB: several meta-blocks (rounds > 1) and a
    second lap of the 8 MiB ring (stale-byte reads past the block end)."""
    data = G.enwik_text(18 << 20, seed=5, vocab=20000)
    got, info = _gpu(ctx, data, 0)
    assert info["rounds"] > 1
    assert got == oracle.encode_plan(data, 5, 22, 0)


def test_rank_pieces_concatenate_to_whole_plan(ctx, oracle):
    """Multi-GPU contract on one GPU: pieces encoded with
    stream_base / is_last equal the slices of the whole plan."""
    data = TEXT4M
    shard, hint = 1 << 17, len(data)
    whole = _oracle_plan(oracle, data, shard)
    q = len(data) // 4
    pieces = [_gpu(ctx, data[a:a + q], shard, hint, stream_base=a, is_last=(a + q == len(data)))[0]
              for a in range(0, len(data), q)]
    assert b"".join(pieces) == whole


def test_unsupported_parameters_fail_loudly(ctx):
    from brotli_amd import hip
    with pytest.raises(hip.BrotliAmdError):
        ctx.encode_host(b"hello world", hip.make_params(11, 22, 0))
    with pytest.raises(hip.BrotliAmdError):
        ctx.encode_host(b"hello world", hip.make_params(10, 22, 0))
    with pytest.raises(hip.BrotliAmdError):
        ctx.encode_host(b"hello world", hip.make_params(5, 30, 0))


def test_full_size_properties(ctx, oracle, ref):
    """BASELINE configs[1] shape at a size the GPU box checks in seconds
    (256 MiB, 256 KiB shards): (1) the stream decodes to the input with the
    reference decoder; (2) a sample of shards is bit-identical to the oracle
    (shards are independent, so shard k of the big job == oracle on slice k);
    (3) re-running is deterministic."""
    import torch
    from brotli_amd import hip
    n, shard = 256 << 20, 1 << 18
    data = G.enwik_text(n)
    p = hip.make_params(5, 22, shard)
    d_in = hip.to_device(data)
    d_out = torch.empty(ctx.max_output(n, p), dtype=torch.uint8, device="cuda:0")
    d_sizes = torch.zeros(n // shard, dtype=torch.int64, device="cuda:0")
    nb, info = ctx.encode_device(d_in, n, p, d_out, d_sizes)
    comp = d_out[:nb].cpu().numpy().tobytes()
    sizes = d_sizes.cpu().numpy()
    assert int(sizes.sum()) == nb and info["nshards"] == n // shard
    assert ref.decompress(comp, n) == data
    offs = np.concatenate(([0], np.cumsum(sizes)))
    for k in (0, 1, 511, 777, n // shard - 1):
        want = oracle.encode_shard(data[k * shard:(k + 1) * shard], 5, 22, min(n, 1 << 30),
                                   min(k * shard, 1 << 30), k == n // shard - 1)
        assert comp[offs[k]:offs[k + 1]] == want, k
    nb2, _ = ctx.encode_device(d_in, n, p, d_out)
    assert nb2 == nb and d_out[:nb].cpu().numpy().tobytes() == comp


# --- quality 1 through the HIP layer (brotli_amd_encode_fast_*) ---------------------------------

def _fast_cases():
    text = G.enwik_text((1 << 20) + 333, seed=13, vocab=20000)
    yield "text", text
    yield "random", G.random_bytes(1 << 20, seed=6)
    yield "mixed", G.mixed_corpus(1 << 20)
    yield "text_rand", text[:300000] + G.random_bytes(200000, seed=8) + text[:150000]
    yield "zeros", bytes(300000)
    yield "rle", (b"abcdefgh" * 50000)[:333333]
    yield "alice", ALICE
    for n in (1, 15, 16, 17, 257, 32768, 32769, 131072, 131073):
        yield "text%d" % n, G.enwik_text(n, seed=n, vocab=2000)


@pytest.mark.parametrize("name,data", list(_fast_cases()))
def test_fast_path_equals_oracle(ctx, oracle, name, data):
    for lgwin in (10, 16, 18, 22):
        if lgwin == 10 and len(data) > 400000:
            continue
        got, nbits, _ = ctx.encode_fast_host(data, lgwin)
        assert nbits % 8 == 0
        assert got == oracle.encode_fast(data, lgwin), (name, lgwin)


@pytest.mark.parametrize("chunk", [65536, 100000, 1 << 19])
def test_fast_path_call_sequences_equal_oracle(ctx, oracle, chunk):
    data = G.enwik_text((2 << 20) + 9, seed=17, vocab=20000) + G.random_bytes(250000, seed=5)
    sizes, off = [], 0
    while off < len(data):
        m = min(chunk, len(data) - off)
        off += m
        sizes.append(m)
    got, _, _ = ctx.encode_fast_host(data, 22, sizes + [0])
    assert got == oracle.encode_fast(data, 22, [(m, 0) for m in sizes] + [(0, 2)])
    # a run that does not end the stream leaves a partial byte for the next run
    got1, nbits, _ = ctx.encode_fast_host(data[:sizes[0]], 22, [sizes[0]], is_last=False)
    carry = (nbits & 7, got1[nbits >> 3] & ((1 << (nbits & 7)) - 1)) if nbits & 7 else (0, 0)
    got2, nbits2, _ = ctx.encode_fast_host(data[sizes[0]:], 22, sizes[1:], is_last=True, carry=carry)
    joined = bytearray(got1[:nbits >> 3]) + got2
    assert bytes(joined) == oracle.encode_fast(data, 22, [(m, 0) for m in sizes[:-1]] + [(sizes[-1], 2)])


def test_fast_path_full_size_properties(ctx, oracle, ref):
    """BASELINE config 3 shape at 256 MiB: quality 1 over random bytes.  Every 4 MiB fragment
    must come out as one raw meta-block (5 header bytes + data), the stream must decode to the
    input, and a 16 MiB prefix must equal the oracle's bytes."""
    import torch
    from brotli_amd import hip
    n = 256 << 20
    g = torch.Generator(device="cuda").manual_seed(1234)
    d_in = torch.zeros(n + hip.INPUT_SLACK, dtype=torch.uint8, device="cuda")
    d_in[:n] = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    d_out = torch.empty(ctx.fast_max_output(n, 1, 22), dtype=torch.uint8, device="cuda")
    nbits, info = ctx.encode_fast_device(d_in, n, d_out, 22)
    assert nbits % 8 == 0
    nfrag = n >> 22
    assert nbits // 8 == 1 + nfrag * ((1 << 22) + 4) + 1 - 0 or nbits // 8 <= n + 8 * nfrag + 8
    got = d_out[:nbits // 8].cpu().numpy().tobytes()
    data = d_in[:n].cpu().numpy().tobytes()
    assert ref.decompress(got, n) == data
    m = 16 << 20
    nb2, _ = ctx.encode_fast_device(d_in, m, d_out, 22)
    assert d_out[:nb2 // 8].cpu().numpy().tobytes() == oracle.encode_fast(data[:m], 22)


def test_parse4_wave_layouts_agree(ctx, oracle, monkeypatch):
    """k_parse4's layouts — 4 / 2 / 1 shards per wave, with and without scout groups — are
    chosen from the shard count; force each one on the same job (BROTLI_AMD_QGROUPS /
    BROTLI_AMD_DUO are read at plan time) and compare with the oracle."""
    from brotli_amd import hip
    data = G.enwik_text((6 << 20) + 4321, seed=23, vocab=30000) + G.mixed_corpus(1 << 20)
    shard = 96 << 10
    want = oracle.encode_plan(data, 5, 22, shard)
    for groups, duo in ((4, 1), (2, 0), (2, 1), (1, 0), (1, 1)):
        monkeypatch.setenv("BROTLI_AMD_QGROUPS", str(groups))
        monkeypatch.setenv("BROTLI_AMD_DUO", str(duo))
        got, _ = ctx.encode_host(data, hip.make_params(5, 22, shard))
        assert got == want, (groups, duo)
