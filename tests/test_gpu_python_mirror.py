"""brotli_amd.brotli mirrors the reference Python module (python/brotli.py,
python/_brotli.c); these tests follow python/tests/compress_test.py and
compressor_test.py (single process, 2 KiB chunks, chunks + flush, concurrent
use) but pin BYTES against the stock library driven with the same calls, and
round-trip through the reference decoder."""
import threading

import pytest

import gen_inputs as G
from test_gpu_abi import _bind, drive, _chunks, ROOT, ALICE
import os

# a kernel that never returns must not take the whole GPU tier with it: pytest-timeout's
# thread method ends the run (a blocked HIP call cannot be interrupted by a signal)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]


@pytest.fixture(scope="module")
def brotli():
    import brotli_amd.brotli as b
    return b


@pytest.fixture(scope="module")
def stock(ref):
    return _bind(os.path.join(ROOT, "oracle", "_ref", "libbrotli_ref.so"))


INPUTS = {"alice29": ALICE, "text300k": G.enwik_text(300000, seed=19, vocab=6000),
          "x": b"x", "empty": b"", "10x10y": b"x" * 10 + b"y" * 10}


@pytest.mark.parametrize("name", list(INPUTS))
def test_compress_equals_reference_calls(brotli, stock, ref, name):
    data = INPUTS[name]
    got = brotli.compress(data, quality=5)
    want, _ = drive(stock, data, [(len(data), 0), (0, 2)])      # process(string) + finish()
    assert got == want
    assert ref.decompress(got, len(data)) == data


@pytest.mark.parametrize("name", ["alice29", "text300k"])
def test_multiple_process(brotli, stock, name):
    data = INPUTS[name]
    c = brotli.Compressor(quality=5)
    out = b"".join(c.process(data[i:i + 2048]) for i in range(0, len(data), 2048)) + c.finish()
    want, _ = drive(stock, data, [(m, 0) for m, _ in _chunks(len(data), 2048, 0)] + [(0, 2)])
    assert out == want


@pytest.mark.parametrize("name", ["alice29"])
def test_multiple_process_and_flush(brotli, stock, ref, name):
    data = INPUTS[name][:40000]
    c = brotli.Compressor(quality=5)
    out = b""
    ops = []
    for i in range(0, len(data), 2048):
        out += c.process(data[i:i + 2048])
        out += c.flush()
        ops += [(len(data[i:i + 2048]), 0), (0, 1)]
    out += c.finish()
    want, _ = drive(stock, data, ops + [(0, 2)])
    assert out == want
    assert ref.decompress(out, len(data)) == data


def test_decompress_round_trip_and_errors(brotli, ref):
    """python/tests/decompress_test.py in small: decompress(compress(x)) == x for the module's own
    output and for the stock encoder's, a damaged or cut stream raises brotli.error."""
    for name, data in INPUTS.items():
        data = bytes(data)
        assert brotli.decompress(brotli.compress(data, quality=5)) == data, name
        assert brotli.decompress(ref.compress(data, 11, 22)) == data, name
    comp = ref.compress(bytes(INPUTS["text300k"]), 5, 22)
    with pytest.raises(brotli.error):
        brotli.decompress(comp[:len(comp) // 2])
    with pytest.raises(brotli.error):
        brotli.decompress(b"\xff" * 100)
    d, got = brotli.Decompressor(), b""
    for i in range(0, len(comp), 20000):
        got += d.process(comp[i:i + 20000])
    assert got == bytes(INPUTS["text300k"]) and d.is_finished()


def test_invalid_arguments_and_unsupported_quality(brotli):
    with pytest.raises(brotli.error):
        brotli.Compressor(quality=12)
    with pytest.raises(brotli.error):
        brotli.Compressor(lgwin=9)
    with pytest.raises(brotli.error):       # valid for the reference, outside the GPU path: loud failure
        brotli.compress(b"hello", quality=11)


def test_use_after_finish(brotli):
    c = brotli.Compressor(quality=5)
    c.process(b"abc")
    c.finish()
    with pytest.raises(brotli.error):
        c.process(b"more")


def test_concurrent_use_raises(brotli):
    """python/tests/compressor_test.py:121-150: sharing one Compressor between
    threads raises brotli.error in at least one of them."""
    c = brotli.Compressor(quality=5)
    data = G.enwik_text(400000, seed=23, vocab=5000)
    errors = []

    def work():
        try:
            for i in range(0, len(data), 50000):
                c.process(data[i:i + 50000])
                c.flush()
        except brotli.error as e:
            errors.append(e)
    ts = [threading.Thread(target=work) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errors


def test_shard_size_extension_equals_oracle_plan(brotli, oracle):
    data = G.enwik_text(1 << 20, seed=11, vocab=20000)
    c = brotli.Compressor(quality=5, size_hint=len(data), shard_size=1 << 17)
    got = c.process(data) + c.finish()
    assert got == oracle.encode_plan(data, 5, 22, 1 << 17)
