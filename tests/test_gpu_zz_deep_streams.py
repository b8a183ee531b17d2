"""Qualities 6-9 as ONE device-resident encoder instance (PROCESS / FLUSH / FINISH sequences,
metadata blocks) through libbrotlienc_amd.so, next to the reference library.

This file sorts last on purpose: the path it drives (k_parse_deep behind the incremental stream
API) was added after the round's GPU budget was spent, so it has run on the host SIMT simulator
(tests/test_sim_kernels.py::test_stream_call_sequences_equal_reference) but not yet on an
MI355X; the per-test limit keeps a surprise from costing the tests before it."""
import ctypes as C

import pytest

import gen_inputs as G
from test_gpu_abi import _chunks, drive, amd, stock  # noqa: F401  (fixtures)

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method="thread")]


@pytest.mark.parametrize("quality,lgwin", [(6, 22), (7, 20), (8, 22), (9, 24)])
def test_deep_quality_stream_sequences_equal_reference(amd, stock, quality, lgwin):
    """Qualities 6-9 as ONE encoder instance with PROCESS / FLUSH / FINISH sequences and a
    metadata block: the device-resident stream runs k_parse_deep and keeps its hash table,
    distance cache and partial byte between calls (streams up to the window size)."""
    data = G.enwik_text(700000, seed=53 + quality, vocab=20000) + G.mixed_corpus(1 << 17)
    params = ((1, quality), (2, lgwin))
    for ops in (_chunks(len(data), 100000, 2, 3), _chunks(len(data), 65536, 2, 0), [(len(data), 2)]):
        got, fin = drive(amd, data, ops, params, take=(len(ops) % 2 == 0))
        want, _ = drive(stock, data, ops, params, take=(len(ops) % 2 == 0))
        assert fin and got == want, (quality, len(ops))
    meta = b"\x01\x02\x03" * 50
    d2 = data[:150000] + meta + data[150000:400000]
    ops = [(150000, 0), (len(meta), 3), (250000, 2)]
    got, fin = drive(amd, d2, ops, params)
    want, _ = drive(stock, d2, ops, params)
    assert fin and got == want


def test_deep_quality_stream_longer_than_window_fails_loudly(amd):
    """k_parse_deep has no ring-wrap rules: a quality-9 stream past the window is refused
    (BROTLI_FALSE), never encoded differently."""
    data = G.enwik_text((1 << 20) + 4096, seed=59, vocab=20000)
    st = amd.BrotliEncoderCreateInstance(None, None, None)
    assert amd.BrotliEncoderSetParameter(st, 1, 9)
    assert amd.BrotliEncoderSetParameter(st, 2, 20)          # window 1 MiB - 16
    buf = C.create_string_buffer(data, len(data))
    n = C.c_size_t(len(data))
    nxt = C.c_void_p(C.addressof(buf))
    ao = C.c_size_t(0)
    no = C.c_void_p(0)
    assert not amd.BrotliEncoderCompressStream(st, 2, C.byref(n), C.byref(nxt), C.byref(ao), C.byref(no), None)
    amd.BrotliEncoderDestroyInstance(st)
