"""The device decoder (brotli_amd/csrc/k_decode.h) on the host SIMT simulator:
the reference ENCODER's output at every quality is the test set (the decoder is written from RFC 7932,
so streams with features our own encoder never emits — qualities 0, 10, 11: all 121 dictionary
transforms, NPOSTFIX / NDIRECT, many block types — are decoded too), the reference DECODER is the
judge of what a valid stream is, and damaged streams must end in an error code, not in a walk
through memory."""
import random

import pytest

import gen_inputs as G
from test_oracle import ALICE


@pytest.fixture(scope="module")
def sim():
    from simharness import Sim
    return Sim()


def _inputs():
    alice = open(ALICE, "rb").read()
    return {"alice": alice[:60000], "text": bytes(G.enwik_text(80000, seed=3, vocab=5000)),
            "mixed": bytes(G.mixed_corpus(120000)), "rand": bytes(G.random_bytes(40000)),
            "zeros": bytes(70000), "tiny": b"x", "empty": b"", "rle": (b"abcdefgh" * 9000)[:70001]}


INPUTS = _inputs()


@pytest.mark.parametrize("name", list(INPUTS))
def test_decodes_reference_streams_at_every_quality(sim, ref, name):
    data = INPUTS[name]
    for quality, lgwin in ((0, 22), (1, 18), (2, 22), (3, 10), (4, 22), (5, 22), (6, 16), (9, 24), (10, 22), (11, 16)):
        if quality >= 10 and len(data) > 70001:
            continue
        comp = ref.compress(data, quality, lgwin)
        # (lane order of the simulator both ways; every third case without the LDS copies of the tables)
        out, res = sim.decode(comp, len(data), reverse=(quality & 1) | (2 if quality % 3 == 0 else 0))
        n, bits, err, fin = res[0]
        assert (err, fin, n) == (0, 1, len(data)) and out == data, (name, quality, lgwin)
        assert (bits + 7) // 8 == len(comp)


def test_font_mode_distance_parameters_and_metadata(sim, ref):
    """NPOSTFIX / NDIRECT != 0 (BROTLI_MODE_FONT, encode.c:616-640) and metadata blocks in the stream."""
    import ctypes as C
    from refharness import PARAM_MODE, PARAM_QUALITY, PARAM_LGWIN
    data = INPUTS["mixed"]
    L = ref.L
    st = L.BrotliEncoderCreateInstance(None, None, None)
    for k, v in ((PARAM_MODE, 2), (PARAM_QUALITY, 6), (PARAM_LGWIN, 20)):
        assert L.BrotliEncoderSetParameter(st, k, v)
    cap = 2 * len(data) + 4096
    out = C.create_string_buffer(cap)
    buf = C.create_string_buffer(data, len(data))
    avail_out, next_out = C.c_size_t(cap), C.c_void_p(C.addressof(out))
    meta = bytes(range(97))
    mbuf = C.create_string_buffer(meta, len(meta))
    for ptr, n, op in ((C.addressof(buf), 50000, 0), (C.addressof(mbuf), len(meta), 3), (C.addressof(buf) + 50000, len(data) - 50000, 2)):
        avail_in, next_in = C.c_size_t(n), C.c_void_p(ptr)
        while True:
            assert L.BrotliEncoderCompressStream(st, op, C.byref(avail_in), C.byref(next_in), C.byref(avail_out),
                                                 C.byref(next_out), None)
            if avail_in.value == 0 and not L.BrotliEncoderHasMoreOutput(st):
                break
    L.BrotliEncoderDestroyInstance(st)
    comp = out.raw[:cap - avail_out.value]
    assert ref.decompress(comp, len(data)) == data
    got, res = sim.decode(comp, len(data))
    assert res[0][2:] == (0, 1) and got == data


@pytest.mark.parametrize("quality,lgwin,shard", [(5, 22, 32768), (9, 18, 50000), (2, 22, 65536), (5, 16, 40000)])
def test_decodes_the_shards_of_a_plan_as_independent_pieces(sim, ref, quality, lgwin, shard):
    """One wave per shard: compressed offsets from the shard sizes, decoded offsets k * shard size,
    only shard 0 carries the stream header, the others are told the window."""
    data = INPUTS["mixed"]
    n, parts, off = len(data), [], 0
    while off < n:
        m = min(shard, n - off)
        parts.append(ref.encode_shard(data[off:off + m], quality, lgwin, n, off, off + m == n))
        off += m
    from brotli_amd.hip import plan_pieces
    pieces = plan_pieces([len(p) for p in parts], n, shard, lgwin)
    out, res = sim.decode(b"".join(parts), n, pieces)
    assert out == data and all(r[2] == 0 for r in res) and res[-1][3] == 1
    assert [r[0] for r in res] == [p[3] for p in pieces]
    # a piece that claims to be isolated but is handed a copy reaching before it: error, not a read
    whole = ref.compress(data, 5, 22)
    _, res = sim.decode(whole, n, [(0, len(whole), 4096, n - 4096, 1 | 2, 22)])
    assert res[0][2] != 0 and res[0][0] <= n - 4096


def test_round_trip_of_the_device_encoder(sim):
    """Encoder kernels -> decoder kernel, both on the simulator, plan and single stream, q1 included."""
    from brotli_amd.hip import plan_pieces
    data = INPUTS["text"]
    for quality in (5, 9, 3):
        comp = sim.encode(data, quality=quality, lgwin=22, shard_size=1 << 15)
        out, res = sim.decode(comp, len(data))          # the concatenation is one valid stream
        assert res[0][2:] == (0, 1) and out == data
    comp = sim.encode_fast(data, lgwin=18)
    out, res = sim.decode(comp, len(data))
    assert res[0][2:] == (0, 1) and out == data


def test_damaged_streams_end_in_an_error(sim, ref):
    data = INPUTS["alice"]
    rng = random.Random(5)
    # a stream that ends in the middle of its last byte with non-zero padding is damaged too
    out, res = sim.decode(b"\xff" * 16, 100)
    assert res[0][2] != 0
    for quality in (5, 11, 1):
        comp = bytearray(ref.compress(data, quality, 22))
        # truncation: never "finished", never more bytes than asked for
        for cut in (1, 2, len(comp) // 3, len(comp) - 1):
            out, res = sim.decode(bytes(comp[:cut]), len(data))
            n, bits, err, fin = res[0]
            assert not (err == 0 and fin == 1) and n <= len(data)
        # bit flips: whatever comes out, the result is an error or a bounded output; the reference
        # decoder is asked the same question and must agree whenever it accepts the stream
        for _ in range(24):
            bad = bytearray(comp)
            p = rng.randrange(len(bad))
            bad[p] ^= 1 << rng.randrange(8)
            out, res = sim.decode(bytes(bad), len(data))
            n, bits, err, fin = res[0]
            assert n <= len(data)
            if err == 0 and fin == 1:
                # what this decoder accepts, the reference accepts, with the same bytes (the converse
                # holds up to trailing data, which the reference's one-shot call lets pass)
                import ctypes as C
                buf = C.create_string_buffer(len(data) + 1)
                sz = C.c_size_t(len(data) + 1)
                assert ref.L.BrotliDecoderDecompress(len(bad), bytes(bad), C.byref(sz), buf) == 1
                assert buf.raw[:sz.value] == out[:n]


def test_python_mirror_decompress_over_the_simulator(sim, ref, monkeypatch):
    """brotli_amd.brotli.decompress / Decompressor (python/brotli.py:56-73, python/_brotli.c:640-930)
    with the device decoder replaced by the simulator: the buffer grows until the stream fits, cut or
    damaged streams and trailing data raise brotli.error, the streaming class hands out the bytes
    once the stream is complete."""
    import brotli_amd.brotli as b

    class OnSim:
        def decode_host(self, comp, n_out, pieces=None, check=True, with_bits=False):
            out, res = sim.decode(comp, n_out)
            return out, [(res[0][0], res[0][2], res[0][3])], [res[0][1]]

    monkeypatch.setattr(b, "_dec_ctx", OnSim())
    data = INPUTS["alice"][:40000]
    comp = ref.compress(data, 5, 22)
    assert b.decompress(comp) == data
    assert b.decompress(ref.compress(bytes(300000), 5, 22)) == bytes(300000)     # 20 bytes -> 300 000
    for bad in (comp[:500], b"\xff" * 100, comp + b"\0"):
        with pytest.raises(b.error):
            b.decompress(bad)
    d, got = b.Decompressor(), b""
    for i in range(0, len(comp), 3000):
        assert not d.is_finished() and d.can_accept_more_data()
        got += d.process(comp[i:i + 3000])
    assert got == data and d.is_finished() and d.process(b"") == b""
    with pytest.raises(b.error):
        d.process(b"x")
    # a first chunk may end anywhere — inside a meta-block header, inside a prefix code: until the stream is
    # complete the reference says "needs more input", whatever the decoder made of the zeroed slack behind the cut
    for quality in (5, 9, 11):
        comp = ref.compress(data, quality, 22)
        for cut in list(range(1, 200, 13)) + [len(comp) // 2, len(comp) - 1]:
            out, res = sim.decode(comp[:cut], len(data))
            assert res[0][2] == 7 and res[0][3] == 0, (quality, cut, res[0])
            d = b.Decompressor()
            assert d.process(comp[:cut]) == b"" and not d.is_finished()
            assert d.process(comp[cut:]) == data and d.is_finished()
    comp = ref.compress(data[:600], 5, 22)
    d, got = b.Decompressor(), b""
    for i in range(len(comp)):                      # byte by byte (every call decodes the prefix again: keep it short)
        got += d.process(comp[i:i + 1])
    assert got == data[:600] and d.is_finished()


def test_damage_never_asks_for_a_bigger_buffer(sim, ref):
    """Error 6 = a meta-block announces more output than the buffer has room for (the Python mirror grows the
    buffer on it); a command that writes past the length its meta-block announced is damage (error 10) and
    must not send a caller on a walk up to 2 GiB buffers."""
    data = INPUTS["alice"][:20000]
    comp = ref.compress(data, 5, 22)
    rng = random.Random(11)
    room = 1 << 20
    for _ in range(150):
        bad = bytearray(comp)
        p = rng.randrange(len(bad))
        bad[p] ^= 1 << rng.randrange(8)
        out, res = sim.decode(bytes(bad), room)
        n, bits, err, fin = res[0]
        assert err != 6 or n + (1 << 24) > room, (p, res[0])      # (a header can announce at most 16 MiB)
    # and the real thing still reports 6
    out, res = sim.decode(comp, 100)
    assert res[0][2] == 6
