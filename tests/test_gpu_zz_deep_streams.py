"""Qualities 6-9 as ONE device-resident encoder instance (PROCESS / FLUSH / FINISH sequences,
metadata blocks) through libbrotlienc_amd.so, next to the reference library.

This file sorts late on purpose: one wave walks one dependency chain here (~1-2 MB/s), these
are the slow tests; the per-test limit keeps a surprise from costing the tests before it."""
import ctypes as C

import pytest

import gen_inputs as G
from test_gpu_abi import _chunks, drive, amd, stock  # noqa: F401  (fixtures)

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method="thread")]


@pytest.mark.parametrize("quality,lgwin", [(6, 22), (7, 20), (8, 22), (9, 24)])
def test_deep_quality_stream_sequences_equal_reference(amd, stock, quality, lgwin):
    """Qualities 6-9 as ONE encoder instance with PROCESS / FLUSH / FINISH sequences and a
    metadata block: the device-resident stream runs k_parse_deep and keeps its hash table,
    distance cache and partial byte between calls."""
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


@pytest.mark.parametrize("quality,lgwin", [(6, 17), (9, 18)])
def test_deep_quality_stream_longer_than_window_equals_reference(amd, stock, quality, lgwin):
    """A quality 6-9 stream several times its window (the ring is lapped, candidates age out of
    the window, matches are not followed across the physical end of the ring,
    hash_longest_match64_inc.h:157-277): one FINISH call and a PROCESS / FLUSH sequence."""
    data = G.enwik_text(1500000, seed=71, vocab=6000)
    params = ((1, quality), (2, lgwin))
    for ops in ([(len(data), 2)], _chunks(len(data), 250000, 2, 2)):
        got, fin = drive(amd, data, ops, params)
        want, _ = drive(stock, data, ops, params)
        assert fin and got == want, (quality, lgwin, len(ops))
