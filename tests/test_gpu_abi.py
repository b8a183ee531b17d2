"""The drop-in boundary on the GPU: the reference's own call sequences driven
through libbrotlienc_amd.so and through the reference library give the same
bytes (same helper, two libraries); the reference CLI linked against our
library compresses alice29.txt to the known answer."""
import ctypes as C
import hashlib
import os
import subprocess

import pytest

import gen_inputs as G

# a kernel that never returns must not take the whole GPU tier with it: pytest-timeout's
# thread method ends the run (a blocked HIP call cannot be interrupted by a signal)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "brotli_amd", "lib")
ALICE = open(os.path.join(ROOT, "tests", "golden", "alice29.txt"), "rb").read()
ALICE_SHA = "b4bf4f4f62af5e94769b822b0f24e4edebdb95477e001b4886eb02f2043834f3"


def _bind(path):
    L = C.CDLL(path)
    L.BrotliEncoderCreateInstance.restype = C.c_void_p
    L.BrotliEncoderCreateInstance.argtypes = [C.c_void_p] * 3
    L.BrotliEncoderDestroyInstance.argtypes = [C.c_void_p]
    L.BrotliEncoderSetParameter.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    L.BrotliEncoderCompressStream.argtypes = [
        C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
        C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.BrotliEncoderIsFinished.argtypes = [C.c_void_p]
    L.BrotliEncoderHasMoreOutput.argtypes = [C.c_void_p]
    L.BrotliEncoderTakeOutput.restype = C.c_void_p
    L.BrotliEncoderTakeOutput.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p,
                                        C.POINTER(C.c_size_t), C.c_char_p]
    L.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
    L.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
    L.BrotliEncoderPrepareDictionary.restype = C.c_void_p
    L.BrotliEncoderPrepareDictionary.argtypes = [C.c_int, C.c_size_t, C.c_char_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
    L.BrotliEncoderDestroyPreparedDictionary.argtypes = [C.c_void_p]
    L.BrotliEncoderAttachPreparedDictionary.argtypes = [C.c_void_p, C.c_void_p]
    if hasattr(L, "BrotliEncoderGetPreparedDictionarySize"):   # (not in the test build of the reference)
        L.BrotliEncoderGetPreparedDictionarySize.restype = C.c_size_t
        L.BrotliEncoderGetPreparedDictionarySize.argtypes = [C.c_void_p]
    return L


@pytest.fixture(scope="module")
def amd():
    return _bind(os.path.join(LIBDIR, "libbrotlienc_amd.so"))


@pytest.fixture(scope="module")
def stock(ref):
    return _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))


def drive(L, data, ops, params=(), out_chunk=1 << 16, take=False, dictionaries=(), attach_before_op=0):
    """ops: list of (nbytes, op).  Returns all output bytes.  `dictionaries`: raw LZ77 prefixes,
    prepared and attached in order (encode.h:318-363) right before operation `attach_before_op`."""
    st = L.BrotliEncoderCreateInstance(None, None, None)
    for k, v in ((1, 5), (2, 22)) + tuple(params):
        assert L.BrotliEncoderSetParameter(st, k, v)
    keep = [C.create_string_buffer(bytes(d), max(len(d), 1)) for d in dictionaries]
    prepared = []

    def attach():
        for d, b in zip(dictionaries, keep):
            pd = L.BrotliEncoderPrepareDictionary(0, len(d), b, 11, None, None, None)
            assert pd
            if hasattr(L, "BrotliEncoderGetPreparedDictionarySize"):
                assert L.BrotliEncoderGetPreparedDictionarySize(pd) > 0
            assert L.BrotliEncoderAttachPreparedDictionary(st, pd)
            prepared.append(pd)
    buf = C.create_string_buffer(bytes(data), len(data))
    out = C.create_string_buffer(out_chunk)
    res = bytearray()
    off = 0
    for k_op, (n, op) in enumerate(ops):
        if k_op == attach_before_op:
            attach()
        avail_in = C.c_size_t(n)
        next_in = C.c_void_p(C.addressof(buf) + off)
        off += n
        while True:
            if take:
                avail_out = C.c_size_t(0)
                next_out = C.c_void_p(0)
            else:
                avail_out = C.c_size_t(out_chunk)
                next_out = C.c_void_p(C.addressof(out))
            assert L.BrotliEncoderCompressStream(st, op, C.byref(avail_in), C.byref(next_in),
                                                 C.byref(avail_out), C.byref(next_out), None)
            took = 0
            if take:
                while True:
                    sz = C.c_size_t(0)
                    p = L.BrotliEncoderTakeOutput(st, C.byref(sz))
                    if not sz.value:
                        break
                    res += C.string_at(p, sz.value)
                    took += sz.value
            else:
                res += out.raw[:out_chunk - avail_out.value]
            # TakeOutput style: the reference does not encode while output is waiting (encode.c:
            # 1694-1697), so a call that handed something over may have left its operation
            # unfinished (at qualities 2 - 3 meta-blocks leave in the middle of a call): the
            # caller repeats the operation until a call yields nothing
            if avail_in.value == 0 and not L.BrotliEncoderHasMoreOutput(st) and took == 0:
                break
    fin = bool(L.BrotliEncoderIsFinished(st))
    L.BrotliEncoderDestroyInstance(st)
    for pd in prepared:
        L.BrotliEncoderDestroyPreparedDictionary(pd)
    return bytes(res), fin


def _chunks(n, size, last_op, flush_every=0):
    ops, off, i = [], 0, 0
    while off < n:
        m = min(size, n - off)
        off += m
        i += 1
        op = last_op if off == n else (1 if flush_every and i % flush_every == 0 else 0)
        ops.append((m, op))
    return ops


def test_one_shot_alice(amd):
    cap = amd.BrotliEncoderMaxCompressedSize(len(ALICE))
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    assert amd.BrotliEncoderCompress(5, 22, 0, len(ALICE), ALICE, C.byref(n), out)
    assert hashlib.sha256(out.raw[:n.value]).hexdigest() == ALICE_SHA


def test_one_shot_empty_and_small_buffer(amd):
    out = C.create_string_buffer(16)
    n = C.c_size_t(16)
    assert amd.BrotliEncoderCompress(5, 22, 0, 0, b"", C.byref(n), out)
    assert out.raw[:n.value] == b"\x06"
    n = C.c_size_t(8)      # too small for alice: FALSE like the reference (encode.c:1340-1353)
    assert not amd.BrotliEncoderCompress(5, 22, 0, len(ALICE), ALICE, C.byref(n), out)


@pytest.mark.parametrize("size,lgwin", [((4 << 20) - 16, 22), ((4 << 20) - 15, 22), (3000000, 22), ((4 << 20) - 16, 24),
                                        ((4 << 20) + 1000, 24), (700000, 20), ((1 << 17) - 16, 17)])
def test_one_shot_quality_5_without_a_plan_equals_reference(amd, stock, monkeypatch, size, lgwin):
    """A stock caller: BrotliEncoderCompress(5, lgwin, ...) and no vendor setting.  An input that fits the window runs
    as a one-shard job on the position index with its chain tiles parsed at once (encode_abi.c submit(), k_chain.h
    tiles) up to 4 MiB; one byte more takes the single-stream path.  Same bytes as the stock library either way."""
    monkeypatch.delenv("BROTLI_AMD_SHARD_KB", raising=False)
    data = G.enwik_text(size, seed=31, vocab=20000)
    outs = []
    for L in (amd, stock):
        cap = L.BrotliEncoderMaxCompressedSize(len(data))
        out = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        assert L.BrotliEncoderCompress(5, lgwin, 0, len(data), data, C.byref(n), out)
        outs.append(out.raw[:n.value])
    assert outs[0] == outs[1]


@pytest.mark.parametrize("quality,lgwin", [(6, 22), (9, 24), (2, 22), (3, 22), (4, 22), (4, 16), (2, 10), (5, 16), (7, 12), (9, 10)])
def test_one_shot_deep_quality_equals_reference(amd, stock, quality, lgwin):
    data = G.enwik_text(1 << 20, seed=29, vocab=10000)
    outs = []
    for L in (amd, stock):
        cap = L.BrotliEncoderMaxCompressedSize(len(data))
        out = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        assert L.BrotliEncoderCompress(quality, lgwin, 0, len(data), data, C.byref(n), out)
        outs.append(out.raw[:n.value])
    assert outs[0] == outs[1]


@pytest.mark.parametrize("quality,lgwin", [(2, 22), (3, 18), (4, 22)])
def test_stream_sequences_at_qualities_2_to_4_equal_reference(amd, stock, quality, lgwin):
    """One encoder instance (the device stream over k_parse_quick) driven with PROCESS / FLUSH /
    FINISH shapes, TakeOutput and a metadata block, against the stock library."""
    data = G.enwik_text(400000, seed=31, vocab=10000)
    params = ((1, quality), (2, lgwin))
    for ops, take in ((_chunks(len(data), 50000, 2), False), (_chunks(len(data), 30000, 2, 3), True),
                      ([(0, 1)] + _chunks(len(data), 150000, 2), False)):
        want, fin_w = drive(stock, data, ops, params, take=take)
        got, fin_g = drive(amd, data, ops, params, take=take)
        assert fin_w and fin_g and got == want, (quality, ops[:3])
    meta = bytes(range(100))
    d2 = data[:100000] + meta + data[100000:300000]
    ops = [(100000, 0), (len(meta), 3), (200000, 2)]
    want, _ = drive(stock, d2, ops, params)
    got, fin = drive(amd, d2, ops, params)
    assert fin and got == want


def test_unsupported_quality_fails_loudly(amd):
    out = C.create_string_buffer(1 << 20)
    n = C.c_size_t(1 << 20)
    assert not amd.BrotliEncoderCompress(11, 22, 0, len(ALICE), ALICE, C.byref(n), out)
    assert n.value == 0


@pytest.mark.parametrize("name,ops_fn", [
    ("single_finish", lambda n: [(n, 2)]),
    ("chunks_2k", lambda n: _chunks(n, 2048, 2)),
    ("chunks_2k_flush_every_16", lambda n: _chunks(n, 2048, 2, 16)),
    ("chunks_100k_flush_each", lambda n: _chunks(n, 100000, 2, 1)),
    ("flush_then_empty_finish", lambda n: [(n, 1), (0, 2)]),
    ("empty_flush_first", lambda n: [(0, 1)] + _chunks(n, 70000, 2)),
])
def test_stream_sequences_equal_reference(amd, stock, name, ops_fn):
    data = G.enwik_text(700000, seed=17, vocab=8000)
    ops = ops_fn(len(data))
    want, fin_w = drive(stock, data, ops)
    got, fin_g = drive(amd, data, ops)
    assert fin_w and fin_g
    assert got == want


def test_take_output_path_equals_reference(amd, stock):
    """Go / Java bindings: available_out = 0 and BrotliEncoderTakeOutput."""
    data = ALICE
    ops = _chunks(len(data), 30000, 2, 2)
    want, _ = drive(stock, data, ops, take=True)
    got, fin = drive(amd, data, ops, take=True)
    assert fin and got == want


def test_stream_offset_parameter(amd, stock):
    data = ALICE[:100000]
    ops = [(len(data), 1)]
    want, _ = drive(stock, data, ops, params=((5, 300000), (9, 200000)))
    got, _ = drive(amd, data, ops, params=((5, 300000), (9, 200000)))
    assert got == want


def test_plan_parameter_equals_oracle_plan(amd, oracle):
    data = G.enwik_text(1 << 20, seed=11, vocab=20000)
    got, fin = drive(amd, data, [(len(data), 2)], params=((5, len(data)), (0x4D490001, 1 << 17)))
    assert fin and got == oracle.encode_plan(data, 5, 22, 1 << 17)


def test_reference_cli_on_our_library():
    """c/tools/brotli.c compiled unmodified and linked against libbrotlienc_amd.so
    (oracle/Makefile target cli_amd): config 1 of BASELINE.json."""
    cli = os.path.join(ROOT, "oracle", "_ref", "brotli_cli_amd")
    if not os.path.exists(cli):
        pytest.skip("oracle/_ref/brotli_cli_amd not built")
    dropin = os.path.join(LIBDIR, "dropin")
    os.makedirs(dropin, exist_ok=True)
    for name, target in (("libbrotlienc.so.1", "../libbrotlienc_amd.so"),
                         ("libbrotli_amd_hip.so", "../libbrotli_amd_hip.so")):
        p = os.path.join(dropin, name)
        if not os.path.lexists(p):
            os.symlink(target, p)
    env = dict(os.environ, LD_LIBRARY_PATH=dropin + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    src = os.path.join(ROOT, "tests", "golden", "alice29.txt")
    r = subprocess.run([cli, "-q", "5", "-w", "22", "-c", src], capture_output=True, env=env, check=True)
    assert hashlib.sha256(r.stdout).hexdigest() == ALICE_SHA
    d = subprocess.run([cli, "-d", "-c"], input=r.stdout, capture_output=True, env=env, check=True)
    assert d.stdout == ALICE


# --- attached dictionaries ----------------------------------------------------

@pytest.mark.parametrize("quality,lgwin", [(5, 22), (5, 18), (6, 22), (7, 20), (9, 24), (4, 22), (3, 18), (2, 22),
                                           (5, 16), (7, 14), (9, 12)])
def test_attached_dictionaries_equal_reference(amd, stock, ref, quality, lgwin):
    """BrotliEncoderPrepareDictionary(RAW) + AttachPreparedDictionary on both libraries (encode.h:
    318-363): one FINISH call, PROCESS / FLUSH shapes with TakeOutput, a dictionary of three chunks,
    a dictionary attached in the middle of the stream, a tiny input against a large dictionary."""
    params = ((1, quality), (2, lgwin))
    for nbytes, dict_bytes, nchunks in ((200000, 80000, 1), (300000, 250000, 3), (3000, 100000, 2)):
        data, chunks = G.dictionary_case(nbytes, dict_bytes, nchunks, seed=quality * 100 + lgwin + nchunks)
        n = len(data)
        for ops, take, at in (([(n, 2)], False, 0), (_chunks(n, 37000, 2, 2), True, 0),
                              ([(n // 3, 1), (n - n // 3, 2)], False, 1)):
            want, fin_w = drive(stock, data, ops, params, take=take, dictionaries=chunks, attach_before_op=at)
            got, fin_g = drive(amd, data, ops, params, take=take, dictionaries=chunks, attach_before_op=at)
            assert fin_w and fin_g and got == want, (quality, lgwin, nbytes, ops[:3], at)
        want, _ = drive(stock, data, [(n, 2)], params, dictionaries=chunks)
        assert ref.decompress_with(want, n, chunks) == data
        if quality != 2:
            plain, _ = drive(stock, data, [(n, 2)], params)
            assert len(want) < len(plain)


def test_golden_vectors_dictionaries(amd):
    """tests/golden/golden.json `dictionary_cases`: sha256 of the reference's output, generated in the
    build container by tests/golden/make_golden.py (the fixture travels, the reference tree does not)."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    for case in gold["dictionary_cases"]:
        data, chunks = G.dictionary_case(**case["input"])
        params = ((1, case["quality"]), (2, case["lgwin"]))
        if case.get("shard_size"):
            params += ((5, len(data)), (0x4D490001, case["shard_size"]))
        got, fin = drive(amd, data, [(len(data), 2)], params, dictionaries=chunks)
        assert fin and len(got) == case["size"] and hashlib.sha256(got).hexdigest() == case["sha256"], case


def test_attached_dictionary_past_one_mebibyte(amd, stock):
    """H68 at quality 5 and H54 at quality 4 (chosen once a MiB is announced); H54 has no dictionary
    variant in the reference and only shifts its distances."""
    data, chunks = G.dictionary_case((1 << 20) + 50000, 150000, 2, seed=77)
    for quality in (5, 4):
        params = ((1, quality), (2, 22))
        want, fin_w = drive(stock, data, [(len(data), 2)], params, dictionaries=chunks)
        got, fin_g = drive(amd, data, [(len(data), 2)], params, dictionaries=chunks)
        assert fin_w and fin_g and got == want, quality


@pytest.mark.parametrize("quality,lgwin", [(5, 22), (9, 22), (3, 18), (6, 14)])
def test_attached_dictionaries_in_a_partition_plan(amd, ref, quality, lgwin):
    """Every shard's instance has the dictionaries attached: bytes equal the reference driven with the
    same plan and Attach calls per instance; one valid stream; the pooled context is clean afterwards."""
    data, chunks = G.dictionary_case(4 << 20, 300000, 2, seed=quality * 10 + lgwin)
    shard = 1 << 16
    want = ref.encode_plan(data, quality, lgwin, shard, dictionaries=chunks)
    assert ref.decompress_with(want, len(data), chunks) == data
    params = ((1, quality), (2, lgwin), (5, len(data)), (0x4D490001, shard))
    got, fin = drive(amd, data, [(len(data), 2)], params, dictionaries=chunks)
    assert fin and got == want
    plain, fin = drive(amd, data, [(len(data), 2)], params)
    assert fin and plain == ref.encode_plan(data, quality, lgwin, shard) and len(want) < len(plain)


def test_dictionary_api_edges(amd):
    """Not-a-dictionary handles, the 15-chunk limit, quality 1 (ignores them)."""
    assert amd.BrotliEncoderPrepareDictionary(1, 4, b"abcd", 11, None, None, None) is None   # serialized: not built
    assert amd.BrotliEncoderGetPreparedDictionarySize(None) == 0
    d = C.create_string_buffer(b"hello hello hello hello", 23)
    pd = amd.BrotliEncoderPrepareDictionary(0, 23, d, 11, None, None, None)
    assert pd and amd.BrotliEncoderGetPreparedDictionarySize(pd) > (1 << 17) * 4
    st = amd.BrotliEncoderCreateInstance(None, None, None)
    for _ in range(15):
        assert amd.BrotliEncoderAttachPreparedDictionary(st, pd)
    assert not amd.BrotliEncoderAttachPreparedDictionary(st, pd)          # SHARED_BROTLI_MAX_COMPOUND_DICTS
    assert not amd.BrotliEncoderAttachPreparedDictionary(st, None)
    amd.BrotliEncoderDestroyInstance(st)
    amd.BrotliEncoderDestroyPreparedDictionary(pd)
    data, chunks = G.dictionary_case(100000, 50000, 1, seed=5)
    with_d, fin = drive(amd, data, [(len(data), 2)], Q1, dictionaries=chunks)
    without, _ = drive(amd, data, [(len(data), 2)], Q1)
    assert fin and with_d == without


def test_reference_cli_with_dictionary_on_our_library(tmp_path):
    """`brotli -D FILE` (c/tools/brotli.c: PrepareDictionary + AttachPreparedDictionary) linked against
    our library writes what the stock build writes, and the stock decoder restores the input."""
    cli, cli_ref = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("brotli_cli_amd", "brotli_cli_ref"))
    if not (os.path.exists(cli) and os.path.exists(cli_ref)):
        pytest.skip("oracle/_ref/brotli_cli_amd / brotli_cli_ref not built")
    dropin = os.path.join(LIBDIR, "dropin")
    os.makedirs(dropin, exist_ok=True)
    for name, target in (("libbrotlienc.so.1", "../libbrotlienc_amd.so"),
                         ("libbrotli_amd_hip.so", "../libbrotli_amd_hip.so")):
        if not os.path.lexists(os.path.join(dropin, name)):
            os.symlink(target, os.path.join(dropin, name))
    env = dict(os.environ, LD_LIBRARY_PATH=dropin + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    data, chunks = G.dictionary_case(400000, 200000, 1, seed=91)
    src, dic = tmp_path / "input.bin", tmp_path / "dictionary.bin"
    src.write_bytes(data)
    dic.write_bytes(chunks[0])
    for q in ("5", "9", "3"):
        args = ["-q", q, "-w", "22", "-D", str(dic), "-c", str(src)]
        got = subprocess.run([cli] + args, capture_output=True, env=env, check=True).stdout
        want = subprocess.run([cli_ref] + args, capture_output=True, check=True).stdout
        assert got == want, q
        back = subprocess.run([cli_ref, "-d", "-D", str(dic), "-c"], input=got, capture_output=True, check=True).stdout
        assert back == data


# --- quality 1 (BrotliEncoderCompressStreamFast / two-pass fragments) ----------

Q1 = ((1, 1),)      # BROTLI_PARAM_QUALITY = 1; `drive` sets quality 5 first, the later value wins


@pytest.mark.parametrize("lgwin", [16, 18, 22, 24])
def test_q1_one_shot_equals_reference(amd, stock, lgwin):
    """BrotliEncoderCompress(quality 1): text, random bytes (BASELINE config 3 in small) and
    a mix; the raw-stream fallback of the one-shot wrapper included (encode.c:1340-1353)."""
    text = G.enwik_text((3 << 20) + 4567, seed=31, vocab=20000)
    for data in (text, G.random_bytes(3 << 20, seed=9), text[:700000] + G.random_bytes(400000, seed=3) + text[:300000],
                 ALICE, b"x", bytes(200000)):
        outs = []
        for L in (amd, stock):
            cap = L.BrotliEncoderMaxCompressedSize(len(data))
            out = C.create_string_buffer(cap)
            n = C.c_size_t(cap)
            assert L.BrotliEncoderCompress(1, lgwin, 0, len(data), data, C.byref(n), out)
            outs.append(out.raw[:n.value])
        assert outs[0] == outs[1], (lgwin, len(data))


@pytest.mark.parametrize("chunk,flush_every,take", [(1 << 19, 0, False), (65536, 3, False), (100000, 0, True),
                                                     (1 << 20, 2, True)])
def test_q1_stream_sequences_equal_reference(amd, stock, chunk, flush_every, take):
    """The CLI's 512 KiB feeds, small feeds with FLUSHes, the Go binding's TakeOutput loop:
    every call is its own run of fragments and the partial byte carries over."""
    data = G.enwik_text((2 << 20) + 77, seed=37, vocab=20000) + G.random_bytes(300000, seed=4)
    ops = _chunks(len(data), chunk, 2, flush_every)
    got, fin = drive(amd, data, ops, Q1, take=take)
    want, _ = drive(stock, data, ops, Q1, take=take)
    assert fin and got == want
    # input size a multiple of the feed: the CLI ends with an empty FINISH call
    n = (len(data) // chunk) * chunk
    ops = [(chunk, 0)] * (n // chunk) + [(0, 2)]
    got, fin = drive(amd, data[:n], ops, Q1)
    want, _ = drive(stock, data[:n], ops, Q1)
    assert fin and got == want


def test_q1_empty_and_flush_only_streams(amd, stock):
    for ops in ([(0, 2)], [(0, 1), (0, 2)], [(0, 1), (5, 1), (0, 1), (0, 2)], [(3, 0), (0, 0), (2, 2)]):
        data = b"hello"[:sum(n for n, _ in ops)]
        got, fin = drive(amd, data, ops, Q1)
        want, _ = drive(stock, data, ops, Q1)
        assert fin and got == want, ops


def test_q1_reference_cli_on_our_library(tmp_path, stock):
    """`brotli -q 1` (the reference CLI built against libbrotlienc_amd.so) = the stock library
    driven the way the CLI drives it: 512 KiB reads, FINISH with the read that hits EOF
    (c/tools/brotli.c:1419-1463)."""
    cli = os.path.join(ROOT, "oracle", "_ref", "brotli_cli_amd")
    if not os.path.exists(cli):
        pytest.skip("oracle/_ref/brotli_cli_amd not built")
    src = tmp_path / "in.bin"
    data = G.enwik_text((1 << 20) + 123, seed=41, vocab=20000)
    src.write_bytes(data)
    dropin = os.path.join(LIBDIR, "dropin")
    env = dict(os.environ, LD_LIBRARY_PATH=dropin + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([cli, "-q", "1", "-w", "22", "-c", str(src)], capture_output=True, env=env, check=True)
    want, _ = drive(stock, data, _chunks(len(data), 1 << 19, 2), Q1)
    assert r.stdout == want


def test_q1_emit_metadata_equals_reference(amd, stock):
    """BROTLI_OPERATION_EMIT_METADATA at quality 1: the metadata header continues the partial
    byte of the data before it (encode.c:1223-1249), also right at the start of the stream and
    with an empty payload; the decoder skips the blocks."""
    from refharness import Ref
    text = G.enwik_text(700000, seed=43, vocab=20000)
    meta = b"metadata payload \x00\x01\x02" * 11
    data = text[:300000] + meta + text[300000:500000] + meta[:1] + text[500000:]
    ops = [(300000, 0), (len(meta), 3), (200000, 1), (0, 3), (1, 3), (200000, 2)]
    got, fin = drive(amd, data, ops, Q1)
    want, _ = drive(stock, data, ops, Q1)
    assert fin and got == want
    assert Ref().decompress(got, len(text)) == text
    ops = [(len(meta), 3), (300000, 2)]
    data = meta + text[:300000]
    got, fin = drive(amd, data, ops, Q1)
    want, _ = drive(stock, data, ops, Q1)
    assert fin and got == want
    # not offered in a partition plan (the shards' partial bytes are sealed by the plan's flushes)
    st = amd.BrotliEncoderCreateInstance(None, None, None)
    amd.BrotliEncoderSetParameter(st, 1, 5)
    amd.BrotliEncoderSetParameter(st, 0x4D490001, 65536)
    n = C.c_size_t(4)
    buf = C.create_string_buffer(b"meta", 4)
    nxt = C.c_void_p(C.addressof(buf))
    ao = C.c_size_t(0)
    no = C.c_void_p(0)
    assert not amd.BrotliEncoderCompressStream(st, 3, C.byref(n), C.byref(nxt), C.byref(ao), C.byref(no), None)
    amd.BrotliEncoderDestroyInstance(st)


def test_q5_single_stream_emit_metadata_equals_reference(amd, stock):
    """EMIT_METADATA on the device-resident quality-5 stream: the pending input is flushed as a
    meta-block without the padding block (encode.c:1569-1573) and the metadata header continues
    its last byte; the hash table, ring positions and distance cache carry on afterwards."""
    from refharness import Ref
    text = G.enwik_text(460000, seed=47, vocab=20000)
    meta = bytes(range(256)) * 3
    for cut in (200000, 5000, 70000):
        data = text[:cut] + meta + text[cut:300000] + text[300000:]
        ops = [(cut, 0), (len(meta), 3), (300000 - cut, 1), (0, 3), (len(text) - 300000, 2)]
        got, fin = drive(amd, data, ops)
        want, _ = drive(stock, data, ops)
        assert fin and got == want, cut
        assert Ref().decompress(got, len(text)) == text
    data = meta[:7] + text[:100000]
    ops = [(7, 3), (100000, 2)]
    got, fin = drive(amd, data, ops)
    want, _ = drive(stock, data, ops)
    assert fin and got == want


# ((5, 22, 23) / (5, 24, 24): blocks longer than the largest chain tile — quality 5 then runs the plain chain over the
#  whole shard, host_plan.h plan_add_tiles: correct and slow)
@pytest.mark.parametrize("quality,lgwin,lgblock", [(5, 22, 17), (5, 18, 20), (9, 22, 16), (6, 20, 21), (4, 22, 18), (5, 22, 12), (5, 22, 23), (5, 24, 24)])
def test_lgblock_parameter_equals_reference(amd, stock, quality, lgwin, lgblock):
    """BROTLI_PARAM_LGBLOCK (encode.h:190-197; quality.h:75-92) on the device: one FINISH (at quality 5 the tiled chain
    with tiles of the caller's block size), a stream fed in pieces with a flush, a partition plan — the stock library's
    bytes, shard by shard for the plan."""
    data = G.enwik_text(1500000, seed=43, vocab=5000) + G.mixed_corpus(300000, seed=9)
    n = len(data)
    params = ((1, quality), (2, lgwin), (5, 1 << 21), (3, lgblock))
    for ops in ([(n, 2)], [(300000, 0), (500001, 1), (n - 800001, 2)]):
        want, fin_w = drive(stock, data, ops, params, out_chunk=1 << 20)
        got, fin_g = drive(amd, data, ops, params, out_chunk=1 << 20)
        assert fin_w and fin_g and got == want, ops
    shard = 400000
    got, fin = drive(amd, data, [(n, 2)], params + ((0x4D490001, shard),), out_chunk=1 << 20)
    parts = []
    for off in range(0, n, shard):
        piece = data[off:off + shard]
        p, f = drive(stock, piece, [(len(piece), 2 if off + shard >= n else 1)], params + ((9, off),) if off else params, out_chunk=1 << 20)
        parts.append(p)
    assert fin and got == b"".join(parts)


def test_golden_vectors_lgblock(amd):
    """tests/golden/golden.json `lgblock_cases`: sha256 of the reference's output with BROTLI_PARAM_LGBLOCK set, from
    tests/golden/make_golden.py (the fixture travels, the reference tree does not).  The size hint is the reference
    plan's: min(total, 1 << 30) for every shard."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    for case in gold["lgblock_cases"]:
        data = G.make(case["input"])
        params = ((1, case["quality"]), (2, case["lgwin"]), (5, len(data)), (3, case["lgblock"]))
        if case.get("shard_size"):
            params += ((0x4D490001, case["shard_size"]),)
        got, fin = drive(amd, data, [(len(data), 2)], params, out_chunk=1 << 20)
        assert fin and len(got) == case["size"] and hashlib.sha256(got).hexdigest() == case["sha256"], case
