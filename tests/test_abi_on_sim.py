"""The drop-in boundary WITHOUT a GPU: brotli_amd/csrc/encode_abi.c built over the host SIMT
simulator (tests/simt/sim_hip_layer.cc -> libbrotlienc_sim.so, test only) and driven with the
reference's call sequences next to the reference library.  The kernels are the shipped
headers; what this adds over tests/test_sim_kernels.py is the host side of the boundary:
parameter latching, size-hint emulation, call lists and the pending partial byte at
quality 1, flush / metadata hand-off, routing to plan / stream jobs."""
import ctypes as C
import hashlib
import os
import subprocess

import pytest

import gen_inputs as G
from refharness import ROOT, TABLES
from test_gpu_abi import ALICE, ALICE_SHA, Q1, _bind, _chunks, drive

SIM_ABI = os.path.join(ROOT, "tests", "simt", "libbrotlienc_sim.so")


@pytest.fixture(scope="module")
def simabi():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "simt")], check=True)
    os.environ["BROTLI_AMD_TABLES"] = TABLES
    return _bind(SIM_ABI)


@pytest.fixture(scope="module")
def stock(ref):
    return _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))


TEXT = G.enwik_text(120000, seed=61, vocab=8000)


def test_one_shot_alice_known_answer(simabi):
    cap = simabi.BrotliEncoderMaxCompressedSize(len(ALICE))
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    assert simabi.BrotliEncoderCompress(5, 22, 0, len(ALICE), ALICE, C.byref(n), out)
    assert hashlib.sha256(out.raw[:n.value]).hexdigest() == ALICE_SHA
    n = C.c_size_t(16)
    assert simabi.BrotliEncoderCompress(5, 22, 0, 0, b"", C.byref(n), out) and out.raw[:n.value] == b"\x06"
    n = C.c_size_t(1 << 20)
    assert not simabi.BrotliEncoderCompress(11, 22, 0, len(ALICE), ALICE, C.byref(n), out) and n.value == 0


@pytest.mark.parametrize("quality,lgwin", [(5, 22), (6, 22), (9, 24), (4, 22), (3, 16), (2, 18), (5, 14), (9, 16)])
def test_single_stream_sequences_and_metadata(simabi, stock, quality, lgwin):
    """One encoder instance: PROCESS / FLUSH / FINISH in assorted shapes, TakeOutput, metadata
    blocks, size hint derived from the calls (no BROTLI_PARAM_SIZE_HINT)."""
    params = ((1, quality), (2, lgwin))
    data = TEXT[:60000]
    for ops, take in ((_chunks(len(data), 2048, 2, 16), False), (_chunks(len(data), 20000, 2, 1), True),
                      ([(0, 1)] + _chunks(len(data), 35000, 2), False), ([(len(data), 1), (0, 2)], False)):
        want, fin_w = drive(stock, data, ops, params, take=take)
        got, fin_g = drive(simabi, data, ops, params, take=take)
        assert fin_w and fin_g and got == want, (quality, ops[:3])
    meta = bytes(range(200))
    d2 = data[:20000] + meta + data[20000:45000] + meta[:1] + data[45000:]
    ops = [(20000, 0), (len(meta), 3), (25000, 1), (0, 3), (1, 3), (15000, 2)]
    want, _ = drive(stock, d2, ops, params, out_chunk=4096)
    got, fin = drive(simabi, d2, ops, params, out_chunk=4096)
    assert fin and got == want
    # one-shot wrapper (quality 6-9: a one-shard job instead of the stream)
    outs = []
    for L in (simabi, stock):
        cap = L.BrotliEncoderMaxCompressedSize(len(data))
        out = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        assert L.BrotliEncoderCompress(quality, lgwin, 0, len(data), data, C.byref(n), out)
        outs.append(out.raw[:n.value])
    assert outs[0] == outs[1]


@pytest.mark.parametrize("quality,lgwin,nbytes,dict_bytes,nchunks,shapes", [
    (5, 18, 100000, 90000, 3, 3), (5, 22, 3000, 100000, 2, 3), (5, 20, 40000, 9, 1, 1),
    (6, 22, 60000, 30000, 2, 2), (9, 24, 60000, 30000, 1, 2), (4, 18, 60000, 30000, 2, 2), (3, 22, 60000, 30000, 1, 2),
    (2, 22, 60000, 30000, 1, 1), (5, 16, 60000, 30000, 2, 2), (9, 12, 60000, 30000, 1, 2)])
def test_attached_dictionaries(simabi, stock, ref, quality, lgwin, nbytes, dict_bytes, nchunks, shapes):
    """BrotliEncoderPrepareDictionary(RAW) + AttachPreparedDictionary next to the reference: the
    index built by dict_index.h, the device lookup (k_dict.h) after every search in k_parse /
    k_parse_deep / k_parse_quick, the gap in the distance codes, ExtendLastCommand into the
    dictionary; one call, a dictionary attached in the middle of the stream, PROCESS / FLUSH with
    TakeOutput."""
    data, chunks = G.dictionary_case(nbytes, dict_bytes, nchunks, seed=quality * 100 + lgwin + nchunks)
    params = ((1, quality), (2, lgwin))
    n = len(data)
    all_shapes = [([(n, 2)], False, 0), ([(n // 3, 1), (n - n // 3, 2)], False, 1), (_chunks(n, 17000, 2, 2), True, 0)]
    for ops, take, at in all_shapes[:shapes]:
        want, fin_w = drive(stock, data, ops, params, take=take, dictionaries=chunks, attach_before_op=at)
        got, fin_g = drive(simabi, data, ops, params, take=take, dictionaries=chunks, attach_before_op=at)
        assert fin_w and fin_g and got == want, (quality, lgwin, ops[:3], at)
    want, _ = drive(stock, data, [(n, 2)], params, dictionaries=chunks)
    assert ref.decompress_with(want, n, chunks) == data
    if dict_bytes > 1000 and quality != 2:
        plain, _ = drive(stock, data, [(n, 2)], params)
        assert len(want) < len(plain)


def test_attached_dictionary_edge_shapes(simabi, stock):
    """Fifteen chunks, chunks too short to have an index entry and an empty one between others,
    BROTLI_PARAM_STREAM_OFFSET, a metadata block in the stream, a dictionary equal to the input (one
    copy through ExtendLastCommand across chunk borders), inputs shorter than a hash, a flush every
    few KB (copies continued into the dictionary at block starts)."""
    data, ch = G.dictionary_case(30000, 20000, 1, seed=1)
    d, n = ch[0], 30000

    def same(ops, params, chunks, data=data, **kw):
        want, fin_w = drive(stock, data, ops, params, dictionaries=chunks, **kw)
        got, fin_g = drive(simabi, data, ops, params, dictionaries=chunks, **kw)
        assert got == want and fin_w == fin_g, (ops[:3], params)

    same([(n, 2)], ((1, 5), (2, 22)), [d[i * 1300:(i + 1) * 1300] for i in range(15)])
    same([(n, 2)], ((1, 5), (2, 22)), [d[:5], d[5:12], b"", d[12:]])
    same([(n, 2)], ((1, 9), (2, 20)), [d[:5], b"", d[5:]])
    same([(n, 1)], ((1, 5), (2, 22), (5, 300000), (9, 100000)), [d])
    same([(n, 2)], ((1, 6), (2, 16), (9, 70000)), [d])
    meta = bytes(range(50))
    same([(10000, 0), (len(meta), 3), (n - 10000, 2)], ((1, 5), (2, 22)), [d],
         data=data[:10000] + meta + data[10000:], out_chunk=4096)
    same([(n, 2)], ((1, 5), (2, 22)), [data[:11000], data[11000:]])
    same([(n, 2)], ((1, 3), (2, 18)), [data])
    same([(7, 2)], ((1, 5), (2, 22)), [d], data=data[:7])
    same([(3, 2)], ((1, 2), (2, 22)), [d], data=data[:3])
    same(_chunks(n, 3000, 2, 1), ((1, 5), (2, 22)), [data[5000:25000]], take=True)
    same(_chunks(n, 5000, 2, 3), ((1, 7), (2, 14)), [data[5000:25000]])
    # attached while input is waiting (found by tools/fuzz_abi_sim.py): the reference has parsed the complete
    # blocks of the PROCESS calls without the dictionary, the partial block and what follows with it
    big, ch2 = G.dictionary_case(100000, 20000, 1, seed=2)
    same([(70000, 0), (30000, 2)], ((1, 5), (2, 18)), ch2, data=big, attach_before_op=1)
    same([(20000, 0), (20000, 0), (60000, 2)], ((1, 3), (2, 14)), ch2, data=big, attach_before_op=2)


@pytest.mark.parametrize("quality,lgwin", [(5, 22), (9, 20), (3, 18), (6, 14)])
def test_attached_dictionaries_in_a_partition_plan(simabi, ref, oracle, quality, lgwin):
    """A plan with dictionaries: every shard's encoder instance has them attached — bytes equal the
    reference driven with the same plan and the same Attach calls on each instance (and the oracle's),
    and the concatenation is one stream that decodes with the dictionary attached once."""
    data, chunks = G.dictionary_case(100000, 60000, 2, seed=quality * 10 + lgwin)
    shard = 1 << 15
    want = ref.encode_plan(data, quality, lgwin, shard, dictionaries=chunks)
    assert ref.decompress_with(want, len(data), chunks) == data
    oracle.set_dictionary(chunks)
    try:
        assert oracle.encode_plan(data, quality, lgwin, shard) == want
    finally:
        oracle.set_dictionary(())
    params = ((1, quality), (2, lgwin), (5, len(data)), (0x4D490001, shard))
    got, fin = drive(simabi, data, [(len(data), 2)], params, dictionaries=chunks)
    assert fin and got == want
    # the next instance on the same (pooled) context starts without them
    plain, fin = drive(simabi, data, [(len(data), 2)], params)
    assert fin and plain == ref.encode_plan(data, quality, lgwin, shard)
    assert len(want) < len(plain)


@pytest.mark.parametrize("groups", [4, 2])
def test_dictionary_plan_on_the_four_shards_per_wave_kernel(simabi, ref, monkeypatch, groups):
    """Quality 5 with dictionaries in a plan runs on k_parse4 since round 6 (compound_lookup16: a 16-lane group per
    shard looks the dictionaries up; until then one shard per wave on k_parse): four and two shards per wave, each
    group at a position of its own — shards of different lengths, three chunks, a copy that runs on into the
    dictionary across a block boundary (ExtendLastCommand's dictionary branch), next to the reference driven with the
    same plan."""
    monkeypatch.setenv("BROTLI_AMD_QGROUPS", str(groups))
    for seed, n, dbytes, nchunks, shard in ((3, 300000, 90000, 3, 40000), (4, 200000, 30000, 1, 1 << 16)):
        data, chunks = G.dictionary_case(n, dbytes, nchunks, seed=seed)
        # a stretch of the first chunk, longer than a block boundary can cut, in the middle of the input
        data = data[:131000] + bytes(chunks[0][100:3100]) + data[131000:]
        want = ref.encode_plan(data, 5, 22, shard, dictionaries=chunks)
        assert ref.decompress_with(want, len(data), chunks) == data
        params = ((1, 5), (2, 22), (5, len(data)), (0x4D490001, shard))
        got, fin = drive(simabi, data, [(len(data), 2)], params, dictionaries=chunks)
        assert fin and got == want, (groups, seed)


def test_large_dictionary_wider_index_and_saturated_keys(simabi, stock):
    """A dictionary past 2 MiB gets a wider index (one more key bit per doubling, compound_dictionary.c:
    163-170) and a run of repeats fills keys past their 32 entries (only the newest 32 stay); the
    dictionary is also far larger than the window of the second case."""
    data, ch = G.dictionary_case(40000, 5 << 20, 1, seed=77)
    d = ch[0][:(5 << 20) - 200000] + b"the quick brown fox " * 10000
    for quality, lgwin in ((5, 22), (3, 16)):
        params = ((1, quality), (2, lgwin))
        want, fin_w = drive(stock, data, [(len(data), 2)], params, dictionaries=[d])
        got, fin_g = drive(simabi, data, [(len(data), 2)], params, dictionaries=[d])
        assert fin_w and fin_g and got == want, (quality, lgwin)


def test_dictionary_api_edges_and_cli(simabi, tmp_path):
    """Not-a-dictionary handles, the 15-chunk limit, quality 1 (ignores them);
    `brotli -D FILE` of the reference CLI over this library (the simulator build) next to the
    stock CLI."""
    L = simabi
    assert L.BrotliEncoderPrepareDictionary(1, 4, b"abcd", 11, None, None, None) is None
    assert L.BrotliEncoderGetPreparedDictionarySize(None) == 0
    d = C.create_string_buffer(b"hello hello hello hello", 23)
    pd = L.BrotliEncoderPrepareDictionary(0, 23, d, 11, None, None, None)
    assert pd and L.BrotliEncoderGetPreparedDictionarySize(pd) > (1 << 17) * 4
    st = L.BrotliEncoderCreateInstance(None, None, None)
    for _ in range(15):
        assert L.BrotliEncoderAttachPreparedDictionary(st, pd)
    assert not L.BrotliEncoderAttachPreparedDictionary(st, pd)
    assert not L.BrotliEncoderAttachPreparedDictionary(st, None)
    L.BrotliEncoderDestroyInstance(st)
    L.BrotliEncoderDestroyPreparedDictionary(pd)
    data, chunks = G.dictionary_case(30000, 20000, 1, seed=5)
    with_d, fin = drive(L, data, [(len(data), 2)], Q1, dictionaries=chunks)
    without, _ = drive(L, data, [(len(data), 2)], Q1)
    assert fin and with_d == without
    cli, cli_ref = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("brotli_cli_amd", "brotli_cli_ref"))
    if not (os.path.exists(cli) and os.path.exists(cli_ref)):
        pytest.skip("oracle/_ref/brotli_cli_amd / brotli_cli_ref not built")
    os.symlink(SIM_ABI, tmp_path / "libbrotlienc.so.1")
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path), BROTLI_AMD_TABLES=TABLES)
    data, chunks = G.dictionary_case(60000, 50000, 1, seed=91)
    (tmp_path / "input.bin").write_bytes(data)
    (tmp_path / "dictionary.bin").write_bytes(chunks[0])
    args = ["-q", "5", "-w", "22", "-D", str(tmp_path / "dictionary.bin"), "-c", str(tmp_path / "input.bin")]
    got = subprocess.run([cli] + args, capture_output=True, env=env, check=True).stdout
    want = subprocess.run([cli_ref] + args, capture_output=True, check=True).stdout
    assert got == want


def test_empty_first_operation_leaves_the_size_hint_open(simabi, stock):
    """An empty EMIT_METADATA / FLUSH before the first data byte must not pin the size hint to 0
    (UpdateSizeHint, encode.c:1619-1632): with >= 1 MiB of data behind it the reference still
    picks the large hasher (H68) at quality 5."""
    # (100 KB of text, repeated: the two hashers already part ways in the first period, and the
    # simulator is through the long copies in seconds)
    data = (G.enwik_text(100000, seed=62, vocab=20000) * 12)[:(1 << 20) + 50000]
    params = ((1, 5), (2, 22))
    ops = [(0, 3), (0, 1), (len(data), 2)]
    want, fin_w = drive(stock, data, ops, params)
    got, fin_g = drive(simabi, data, ops, params)
    assert fin_w and fin_g and got == want


def test_stream_offset_and_plan_parameters(simabi, stock, oracle):
    data = ALICE[:100000]
    ops = [(len(data), 1)]
    want, _ = drive(stock, data, ops, params=((5, 300000), (9, 200000)))
    got, _ = drive(simabi, data, ops, params=((5, 300000), (9, 200000)))
    assert got == want
    got, fin = drive(simabi, TEXT, [(len(TEXT), 2)], params=((5, len(TEXT)), (0x4D490001, 1 << 15)))
    assert fin and got == oracle.encode_plan(TEXT, 5, 22, 1 << 15)
    # a FLUSH ends the current shards, the next input starts new ones at its stream offset
    ops = [(50000, 1), (70000, 2)]
    got, fin = drive(simabi, TEXT, ops, params=((5, len(TEXT)), (0x4D490001, 1 << 15)))
    parts, off = [], 0
    for n_call, op in ops:
        o2 = 0
        while o2 < n_call:
            m = min(1 << 15, n_call - o2)
            parts.append(oracle.encode_shard(TEXT[off + o2:off + o2 + m], 5, 22, len(TEXT), off + o2,
                                             op == 2 and o2 + m == n_call))
            o2 += m
        off += n_call
    assert fin and got == b"".join(parts)


def test_process_calls_reach_the_device_before_the_final_operation(simabi, stock, oracle, monkeypatch):
    """PROCESS forwards what is waiting once it passes BROTLI_AMD_FEED_KB (bounded host memory,
    output during PROCESS, encode.c:1665-1722): the bytes are those of the same calls without
    forwarding — the reference's for one instance, the plan's for a partition plan."""
    monkeypatch.setenv("BROTLI_AMD_FEED_KB", "100")
    data = G.enwik_text(223457, seed=63, vocab=20000)
    ops = _chunks(len(data), 50000, 2)
    # one encoder instance (quality 5): forwarded to the device stream with OP_PROCESS
    want, _ = drive(stock, data, ops, ((1, 5), (2, 22)))
    got, fin = drive(simabi, data, ops, ((1, 5), (2, 22)), out_chunk=1 << 15)
    assert fin and got == want
    # partition plan of 64 KiB shards: complete shards leave during PROCESS
    shard = 1 << 16
    got, fin = drive(simabi, data, ops, ((1, 5), (2, 22), (0x4D490001, shard)))
    parts, off = [], 0
    # (the size hint is what the calls had shown when the first input block filled: the first call)
    while off < len(data):
        m = min(shard, len(data) - off)
        parts.append(oracle.encode_shard(data[off:off + m], 5, 22, 50000, off, off + m == len(data)))
        off += m
    assert fin and got == b"".join(parts)


def test_quality_1_call_patterns_and_metadata(simabi, stock):
    data = TEXT + G.random_bytes(150000, seed=4) + TEXT[:80000]
    for ops, take in ((_chunks(len(data), 1 << 17, 2), False), (_chunks(len(data), 65536, 2, 3), False),
                      (_chunks(len(data), 100000, 2), True), ([(80000, 0)] * 4 + [(0, 2)], False),
                      ([(0, 2)], False), ([(0, 1), (5, 1), (0, 1), (0, 2)], False)):
        d = data[:sum(n for n, _ in ops)]
        want, _ = drive(stock, d, ops, Q1, take=take)
        got, fin = drive(simabi, d, ops, Q1, take=take)
        assert fin and got == want, ops[:3]
    meta = b"metadata payload \x00\x01\x02" * 11
    d2 = data[:120000] + meta + data[120000:200000] + meta[:1] + data[200000:260000]
    ops = [(120000, 0), (len(meta), 3), (80000, 1), (0, 3), (1, 3), (60000, 2)]
    want, _ = drive(stock, d2, ops, Q1, out_chunk=8192)
    got, fin = drive(simabi, d2, ops, Q1, out_chunk=8192)
    assert fin and got == want
    for lgwin in (16, 22):
        outs = []
        for L in (simabi, stock):
            cap = L.BrotliEncoderMaxCompressedSize(len(data))
            out = C.create_string_buffer(cap)
            n = C.c_size_t(cap)
            assert L.BrotliEncoderCompress(1, lgwin, 0, len(data), data, C.byref(n), out)
            outs.append(out.raw[:n.value])
        assert outs[0] == outs[1]


@pytest.mark.parametrize("quality,lgwin", [(9, 17)])
def test_deep_stream_longer_than_its_window(simabi, stock, quality, lgwin):
    """Qualities 6-9 on one stream several windows long: candidates age out of the window, the
    ring is lapped (stale byte behind a block end), matches are not followed across the physical
    end of the ring (hash_longest_match64_inc.h:187-195, 243-249) — and after a FLUSH the input
    blocks are no longer aligned to the ring, so a block crosses that end."""
    data = G.enwik_text(330000, seed=71, vocab=6000)
    params = ((1, quality), (2, lgwin))
    for ops in ([(len(data), 2)], _chunks(len(data), 120000, 2, 2)):
        want, _ = drive(stock, data, ops, params)
        got, fin = drive(simabi, data, ops, params)
        assert fin and got == want, (quality, lgwin, len(ops))


def test_boundary_error_behaviour(simabi):
    st = simabi.BrotliEncoderCreateInstance(None, None, None)
    assert simabi.BrotliEncoderSetParameter(st, 1, 11) and simabi.BrotliEncoderSetParameter(st, 2, 20)
    data = G.enwik_text(5000, seed=59, vocab=2000)                  # quality 11: outside the GPU path
    buf = C.create_string_buffer(data, len(data))
    n = C.c_size_t(len(data))
    nxt = C.c_void_p(C.addressof(buf))
    ao = C.c_size_t(0)
    no = C.c_void_p(0)
    assert not simabi.BrotliEncoderCompressStream(st, 2, C.byref(n), C.byref(nxt), C.byref(ao), C.byref(no), None)
    assert not simabi.BrotliEncoderSetParameter(st, 1, 5)           # parameters are latched (encode.c:63)
    simabi.BrotliEncoderDestroyInstance(st)
    st = simabi.BrotliEncoderCreateInstance(None, None, None)
    simabi.BrotliEncoderSetParameter(st, 1, 5)
    simabi.BrotliEncoderSetParameter(st, 0x4D490001, 65536)
    n = C.c_size_t(4)
    b4 = C.create_string_buffer(b"meta", 4)
    nxt = C.c_void_p(C.addressof(b4))
    assert not simabi.BrotliEncoderCompressStream(st, 3, C.byref(n), C.byref(nxt), C.byref(ao), C.byref(no), None)
    simabi.BrotliEncoderDestroyInstance(st)


def test_python_mirror_over_the_sim_library(simabi, stock, monkeypatch):
    """brotli_amd.brotli (the mirror of python/brotli.py) bound to the simulator-backed library:
    compress(), Compressor.process / flush / finish at qualities 1, 5 and 9, its errors."""
    import brotli_amd.brotli as b
    monkeypatch.setattr(b, "_LIB_PATH", SIM_ABI)
    monkeypatch.setattr(b, "_lib", None)
    data = TEXT[:60000]
    for quality in (1, 5, 9):
        got = b.compress(data, quality=quality, lgwin=22)
        want, _ = drive(stock, data, [(len(data), 0), (0, 2)], ((1, quality), (2, 22)))      # process(string) + finish()
        assert got == want, quality
        c = b.Compressor(quality=quality, lgwin=22)
        out, ops = b"", []
        for i in range(0, len(data), 16384):
            out += c.process(data[i:i + 16384])
            ops.append((len(data[i:i + 16384]), 0))
            if (i // 16384) % 2:
                out += c.flush()
                ops.append((0, 1))
        out += c.finish()
        ops.append((0, 2))
        want, _ = drive(stock, data, ops, ((1, quality), (2, 22)))
        assert out == want, quality
        with pytest.raises(b.error):
            c.process(b"more")
    with pytest.raises(b.error):
        b.compress(data, quality=11)          # outside the GPU path: fails loudly
    monkeypatch.setattr(b, "_lib", None)


@pytest.mark.parametrize("quality,lgwin", [(5, 22), (9, 22), (7, 18), (3, 22)])
def test_disable_literal_context_modeling(simabi, stock, quality, lgwin):
    """BROTLI_PARAM_DISABLE_LITERAL_CONTEXT_MODELING (encode.h:194-201, encode.c:561): the build kernel keeps one literal
    context; a stream in two flushes, a partition plan, and the one-shot call give the stock library's bytes — and,
    at quality >= 5 on text, not the bytes of the default."""
    data = G.enwik_text(150000, seed=41, vocab=3000)
    n = len(data)
    params = ((1, quality), (2, lgwin), (5, 1 << 20), (4, 1))
    for ops in ([(n, 2)], [(n // 2, 1), (n - n // 2, 2)]):
        want, fin_w = drive(stock, data, ops, params)
        got, fin_g = drive(simabi, data, ops, params)
        assert fin_w and fin_g and got == want
    if quality >= 5:
        default, _ = drive(stock, data, [(n, 2)], params[:3])
        assert default != want
    got, fin = drive(simabi, data, [(n, 2)], params + ((0x4D490001, 1 << 16),))
    parts = []
    for off in range(0, n, 1 << 16):
        piece = data[off:off + (1 << 16)]
        p, f = drive(stock, piece, [(len(piece), 2 if off + (1 << 16) >= n else 1)], params + ((9, off),) if off else params)
        parts.append(p)
    assert fin and got == b"".join(parts)


@pytest.mark.parametrize("quality,lgwin,lgblock", [(5, 22, 17), (5, 18, 20), (9, 22, 16), (4, 22, 18), (7, 16, 24), (3, 20, 20)])
def test_lgblock_parameter(simabi, stock, quality, lgwin, lgblock):
    """BROTLI_PARAM_LGBLOCK (encode.h:190-197; ComputeLgBlock, quality.h:75-92: looked at from quality 4 on, clamped to
    16 .. 24): the input block size changes where matches are cut, what is stitched, the ring and the largest
    meta-block.  One FINISH, a stream fed in pieces with a flush, and a partition plan give the stock library's bytes —
    and, where the parameter counts and differs from the default, not the default's."""
    data = G.enwik_text(420000, seed=43, vocab=5000) + G.mixed_corpus(150000, seed=9)
    n = len(data)
    params = ((1, quality), (2, lgwin), (5, 1 << 20), (3, lgblock))
    for ops in ([(n, 2)], [(100000, 0), (150001, 1), (n - 250001, 2)]):
        want, fin_w = drive(stock, data, ops, params)
        got, fin_g = drive(simabi, data, ops, params)
        assert fin_w and fin_g and got == want, ops
    default_lg = 14 if quality < 4 else (min(18, lgwin) if quality >= 9 and lgwin > 16 else 16)
    eff = default_lg if quality < 4 else min(24, max(16, lgblock))
    if eff != default_lg:
        default, _ = drive(stock, data, [(n, 2)], params[:3])
        assert default != want
    shard = 200000
    got, fin = drive(simabi, data, [(n, 2)], params + ((0x4D490001, shard),))
    parts = []
    for off in range(0, n, shard):
        piece = data[off:off + shard]
        p, f = drive(stock, piece, [(len(piece), 2 if off + shard >= n else 1)], params + ((9, off),) if off else params)
        parts.append(p)
    assert fin and got == b"".join(parts)
